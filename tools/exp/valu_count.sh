#!/bin/bash
# dynamic VALU / SALU / LDS wave-instructions per wave of the kernels tools/exp/rates.py runs.  usage: bash tools/exp/valu_count.sh c3
export TMPDIR=/tmp
P=/tmp/vc; rm -rf $P; mkdir -p $P
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES \
   --output-format csv -d $P/a -- python3 tools/exp/rates.py "$@" > /dev/null 2>&1
python3 tools/sq_summary.py $P/a/*/*counter_collection.csv | grep -E "^mapf|per wave|VALU active"
