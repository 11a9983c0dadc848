#!/bin/bash
# A/B of the C5 share (16384 envs x 32 agents) and C5 whole on ONE box: bench.py under each library given.
#   gpurun -- 'bash tools/exp/c5_ab.sh 2 default variants/libmapf_hip_r4c.so'
N=${1:-2}; shift || true
LIBDIR=$PWD/gym-mapf_amd/gym_mapf_amd/lib
for i in $(seq $N); do
  for lib in "$@"; do
    if [ "$lib" = default ]; then unset MAPF_HIP_LIB; else export MAPF_HIP_LIB=$LIBDIR/$lib; fi
    for cfgflags in "--config c5 --envs 16384" "--config c5"; do
      echo -n "[$lib] [$cfgflags] "
      python3 bench.py $cfgflags --steps 10 --warmup 3 --repeats 3 --no-side-legs --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f G  frac %.4f  %s' % (d['value']/1e9, d['roofline']['frac'], d['roofline']['kernel'][:100]))"
    done
  done
done
