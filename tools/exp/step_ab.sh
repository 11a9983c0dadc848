#!/bin/bash
# A/B of the single step on ONE box: every library given (paths relative to gym_mapf_amd/lib; "default" = the shipped one)
# runs tools/single_step_scaling.py's child mode at the given batch, interleaved, N rounds.
#   gpurun -- 'bash tools/exp/step_ab.sh 65536 3 default variants/libmapf_hip_r4a.so'
E=${1:-65536}; N=${2:-3}; NODES=${NODES:-16}; shift 2 || true
LIBDIR=$PWD/gym-mapf_amd/gym_mapf_amd/lib
for i in $(seq $N); do
  for lib in "$@"; do
    if [ "$lib" = default ]; then unset MAPF_HIP_LIB; else export MAPF_HIP_LIB=$LIBDIR/$lib; fi
    echo -n "[$lib] "
    python3 tools/single_step_scaling.py $E 16 graph $NODES 2>/dev/null | tail -1 | cut -c1-150
  done
done
