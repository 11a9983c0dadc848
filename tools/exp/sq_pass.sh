#!/bin/bash
# SQ counter passes over tools/exp/rates.py <cases...>; summary to stdout.   usage: bash tools/exp/sq_pass.sh c3 c4s_quad
export TMPDIR=/tmp
P=/tmp/sqp; rm -rf $P; mkdir -p $P
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE \
   --output-format csv -d $P/a -- python3 tools/exp/rates.py "$@" > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR \
   --output-format csv -d $P/b -- python3 tools/exp/rates.py "$@" > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_INSTS_BRANCH \
   --output-format csv -d $P/c -- python3 tools/exp/rates.py "$@" > /dev/null 2>&1
python3 tools/sq_summary.py $P/a/*/*counter_collection.csv $P/b/*/*counter_collection.csv $P/c/*/*counter_collection.csv
