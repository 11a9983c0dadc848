#!/bin/bash
# Kernel durations, SQ counters and HBM write bytes of the env.P kernel (mapf_transitions) on one configuration:
#   gpurun -- 'bash tools/exp/transitions_prof.sh 8 20000 [compact] > gpurun_out/transitions_prof_a8.txt'
set -eo pipefail
export TMPDIR=/tmp
A=$1; N=$2; MODE=${3:-reserved}
P=/tmp/trprof; rm -rf $P; mkdir -p $P
CMD="python3 tools/prof_transitions.py $A $N 10 $MODE"
rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats -- $CMD
echo "--- kernel stats (rocprofv3 --kernel-trace --stats)"
cat $P/stats/*/*kernel_stats.csv
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $P/sq1 -- $CMD > /dev/null
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR --output-format csv -d $P/sq2 -- $CMD > /dev/null
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 --output-format csv -d $P/sq3 -- $CMD > /dev/null
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $P/wr -- $CMD > /dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $P/rd -- $CMD > /dev/null
echo "--- SQ counters (three passes), per dispatch"
python3 tools/sq_summary.py --match transitions $P/sq1/*/*counter_collection.csv $P/sq2/*/*counter_collection.csv $P/sq3/*/*counter_collection.csv
echo "--- HBM bytes per dispatch (WRITE_SIZE, FETCH_SIZE in KiB; separate passes)"
python3 tools/sq_summary.py --match transitions $P/wr/*/*counter_collection.csv $P/rd/*/*counter_collection.csv
