#!/bin/bash
# SQ counter passes (sq1 + sq2 + sq3 of tools/refresh_profiles.sh) of one bench.py command line; summary on stdout.
#   gpurun -- 'MAPF_TUNE=k=4 bash tools/exp/sq_pass.sh --envs 131072 --no-side-legs'
set -eo pipefail
export TMPDIR=/tmp
P=/tmp/sqpass; rm -rf $P; mkdir -p $P
BENCH="python3 bench.py --steps 10 --warmup 3 --repeats 2 --no-cpu-baseline --no-scalar-env --no-per-gpu-shapes --no-policy-rollout --no-transitions $*"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $P/sq1 -- $BENCH > /dev/null
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR --output-format csv -d $P/sq2 -- $BENCH > /dev/null
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 --output-format csv -d $P/sq3 -- $BENCH > /dev/null
python3 tools/sq_summary.py $P/sq1/*/*counter_collection.csv $P/sq2/*/*counter_collection.csv $P/sq3/*/*counter_collection.csv
python3 tools/kernel_durations.py $P/sq1/*/*kernel_trace.csv
