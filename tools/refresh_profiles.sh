#!/bin/bash
# Regenerates the measurement artefacts kept under profiles/ on a GPU box:
#   gpurun --timeout 1200 -- 'bash tools/refresh_profiles.sh r05 c3 c3p c4s'
#   gpurun --timeout 1200 -- 'bash tools/refresh_profiles.sh r05 c5s c5'
# writes gpurun_out/prof_<tag>/<tag>_<config>_*; copy them into profiles/ afterwards (gpurun_out/ is scratch) and
# run tools/derive_traffic.py / tools/derive_valu.py there is no need: the script does it and copies the two JSON files
# next to its other outputs.  Every rocprofv3 call has the program itself right after `--`; PMC passes use
# --kernel-trace only.  Configurations: c3 = the bench default (configs[2]); c4s = configs[3]'s share of one GPU
# (32768 envs); c5s = configs[4]'s share (16384 envs x 32 agents); c5 = configs[4] whole on one GPU; c3p = c3 with the in-kernel
# policy stream (actions = NULL: the policy_rollout leg's kernel).
set -eo pipefail
tag=${1:-r05}
shift || true
configs=${*:-c3}
out=$PWD/gpurun_out/prof_$tag
mkdir -p "$out"
# the static instruction mix (tools/valu_mix.py, made in the build container) must belong to THESE kernel sources
python3 -c "import json, bench; assert json.load(open('profiles/valu_mix.json'))['csrc_hash'] == bench.csrc_hash(), 'run tools/valu_mix.py first'"
export TMPDIR=/tmp
P=/tmp/prof
rm -rf $P && mkdir -p $P

for cfg in $configs; do
  case $cfg in
    c3)  flags=""; E=65536; A=8 ;;
    c3p) flags="--policy-actions"; E=65536; A=8 ;;
    c4s) flags="--config c4 --envs 32768"; E=32768; A=8 ;;
    c5s) flags="--config c5 --envs 16384"; E=16384; A=32 ;;
    c5)  flags="--config c5"; E=131072; A=32 ;;
    *) echo "unknown config $cfg"; exit 1 ;;
  esac
  BENCH="python3 bench.py --steps 20 --warmup 5 --repeats 2 --no-cpu-baseline --no-scalar-env --no-per-gpu-shapes --no-policy-rollout --no-transitions $flags"
  pre="$out/${tag}_${cfg}"

  # 1. the command under the profiler: per-kernel durations
  rocprofv3 --kernel-trace --stats --output-format csv -d $P/$cfg/stats -- $BENCH > "${pre}_bench_under_rocprof.json"
  cp $P/$cfg/stats/*/*kernel_stats.csv "${pre}_bench_kernel_stats.csv"
  python3 tools/kernel_durations.py $P/$cfg/stats/*/*kernel_trace.csv > "${pre}_bench_kernel_durations.txt"
  echo "$cfg: stats done"

  # 2. HBM traffic: FETCH_SIZE and WRITE_SIZE in separate passes, at two launch lengths (per-step + fixed part)
  for T in 256 128; do
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $P/$cfg/fetch$T -- $BENCH --rollout-steps $T > /dev/null
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $P/$cfg/write$T -- $BENCH --rollout-steps $T > /dev/null
    python3 tools/pmc_compact.py $P/$cfg/fetch$T/*/*counter_collection.csv > "${pre}_pmc_fetch_size_T$T.csv"
    python3 tools/pmc_compact.py $P/$cfg/write$T/*/*counter_collection.csv > "${pre}_pmc_write_size_T$T.csv"
  done
  label=$(python3 -c "import json,sys; d=json.load(open('${pre}_bench_under_rocprof.json')); print(d['roofline']['kernel'] + '||' + d['single_step_launches']['kernel'])")
  echo "$label" > $P/$cfg/label
  echo "$cfg: traffic passes done"

  # 3. SQ counters of the same command (8 SQ slots + GRBM per pass): what bounds the kernels
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE \
            --output-format csv -d $P/$cfg/sq1 -- $BENCH > /dev/null
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR \
            --output-format csv -d $P/$cfg/sq2 -- $BENCH > /dev/null
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 \
            --output-format csv -d $P/$cfg/sq3 -- $BENCH > /dev/null
  for k in 1 2 3; do python3 tools/pmc_compact.py $P/$cfg/sq$k/*/*counter_collection.csv > "${pre}_pmc_sq_pass$k.csv"; done
  python3 tools/sq_summary.py $P/$cfg/sq1/*/*counter_collection.csv $P/$cfg/sq2/*/*counter_collection.csv $P/$cfg/sq3/*/*counter_collection.csv > "${pre}_sq_counters_summary.txt"
  python3 tools/derive_valu.py "${label%%||*}" $E $A 256 $P/$cfg/sq1/*/*counter_collection.csv $P/$cfg/sq1/*/*kernel_trace.csv $P/$cfg/sq3/*/*counter_collection.csv > /dev/null
  echo "$cfg: sq done"
done

# traffic.json is written whole from all the configurations of this call PLUS what profiles/traffic.json already holds
# for other batches (derive_traffic.py --merge)
args=()
for cfg in $configs; do
  case $cfg in c3|c3p) E=65536; A=8 ;; c4s) E=32768; A=8 ;; c5s) E=16384; A=32 ;; c5) E=131072; A=32 ;; esac
  args+=("$(cat $P/$cfg/label)" $E $A 256 $P/$cfg/fetch256/*/*counter_collection.csv $P/$cfg/write256/*/*counter_collection.csv \
         128 $P/$cfg/fetch128/*/*counter_collection.csv $P/$cfg/write128/*/*counter_collection.csv)
done
python3 tools/derive_traffic.py --merge "${args[@]}" > /dev/null
cp profiles/traffic.json "$out/traffic.json"
cp profiles/valu.json "$out/valu.json"
echo "traffic + valu done"

# 4. the un-profiled lines, AFTER traffic.json / valu.json exist (so that `roofline.traffic` / `valu_frac` are live in them --
# the line under rocprofv3 above was printed before the counters of these sources had been derived): every configuration's
# bench line; for c3 also the driver's exact command and the default
for cfg in $configs; do
  case $cfg in
    c3)  flags="" ;;
    c3p) flags="--policy-actions" ;;
    c4s) flags="--config c4 --envs 32768" ;;
    c5s) flags="--config c5 --envs 16384" ;;
    c5)  flags="--config c5" ;;
  esac
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-scalar-env $flags > "$out/${tag}_${cfg}_bench.json"
  if [ "$cfg" = c3 ]; then
    python3 tools/exp/launch_series.py 300 > "$out/${tag}_launch_series.txt" 2>&1
    python3 bench.py --steps 20 --warmup 5 > "$out/${tag}_bench_steps20_warmup5.json"
    python3 bench.py > "$out/${tag}_bench_default.json"
    cat "$out/${tag}_bench_default.json"
  fi
done
