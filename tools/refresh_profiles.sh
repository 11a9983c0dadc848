#!/bin/bash
# Regenerates the measurement artefacts kept under profiles/ on a GPU box:
#   gpurun --timeout 1200 -- 'bash tools/refresh_profiles.sh r02'
# writes gpurun_out/<tag>_*; copy them into profiles/ afterwards (gpurun_out/ is scratch).
# Every rocprofv3 call has the program itself right after `--`; PMC passes use --kernel-trace only.
set -eo pipefail
tag=${1:-r02}
out=$PWD/gpurun_out/prof_$tag
mkdir -p "$out"
export TMPDIR=/tmp
P=/tmp/prof
rm -rf $P && mkdir -p $P
BENCH="python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-scalar-env"

# 1. the driver's command under the profiler: per-kernel durations
rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats -- $BENCH > "$out/${tag}_bench_under_rocprof.json"
cp $P/stats/*/*kernel_stats.csv "$out/${tag}_bench_kernel_stats.csv"
python3 tools/kernel_durations.py $P/stats/*/*kernel_trace.csv > "$out/${tag}_bench_kernel_durations.txt"
echo "stats done"

# 2. HBM traffic: FETCH_SIZE and WRITE_SIZE in separate passes, at two launch lengths (per-step + fixed part)
for T in 256 128; do
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $P/fetch$T -- $BENCH --rollout-steps $T > /dev/null
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $P/write$T -- $BENCH --rollout-steps $T > /dev/null
  python3 tools/pmc_compact.py $P/fetch$T/*/*counter_collection.csv > "$out/${tag}_pmc_fetch_size_T$T.csv"
  python3 tools/pmc_compact.py $P/write$T/*/*counter_collection.csv > "$out/${tag}_pmc_write_size_T$T.csv"
done
label=$(python3 -c "import json,sys; d=json.load(open('$out/${tag}_bench_under_rocprof.json')); print(d['roofline']['kernel'] + '||' + d['single_step_launches']['kernel'])")
python3 tools/derive_traffic.py "$label" 65536 8 256 $P/fetch256/*/*counter_collection.csv $P/write256/*/*counter_collection.csv \
        128 $P/fetch128/*/*counter_collection.csv $P/write128/*/*counter_collection.csv > /dev/null
cp profiles/traffic.json "$out/traffic.json"
echo "traffic done"

# 3. SQ counters of the same command (8 SQ slots + GRBM per pass): what bounds the kernels
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE \
          --output-format csv -d $P/sq1 -- $BENCH > /dev/null
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR \
          --output-format csv -d $P/sq2 -- $BENCH > /dev/null
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 \
          --output-format csv -d $P/sq3 -- $BENCH > /dev/null
for k in 1 2 3; do python3 tools/pmc_compact.py $P/sq$k/*/*counter_collection.csv > "$out/${tag}_pmc_sq_pass$k.csv"; done
python3 tools/sq_summary.py $P/sq1/*/*counter_collection.csv $P/sq2/*/*counter_collection.csv $P/sq3/*/*counter_collection.csv > "$out/${tag}_sq_counters_summary.txt"
echo "sq done"

# 4. the un-profiled lines: the driver's command and the default
python3 tools/exp/launch_series.py 300 > "$out/${tag}_launch_series.txt" 2>&1
python3 bench.py --steps 20 --warmup 5 > "$out/${tag}_bench_steps20_warmup5.json"
python3 bench.py > "$out/${tag}_bench_default.json"
cat "$out/${tag}_bench_default.json"
