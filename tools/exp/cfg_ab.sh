#!/bin/bash
# A/B of bench.py configurations on ONE box: every library given runs every "flags" string, interleaved, N rounds.
#   gpurun -- 'bash tools/exp/cfg_ab.sh 2 "--config c2|--config c4 --envs 32768" default variants/libmapf_hip_r3.so'
N=${1:-2}; IFS='|' read -ra FLAGS <<< "$2"; shift 2 || true
LIBDIR=$PWD/gym-mapf_amd/gym_mapf_amd/lib
for i in $(seq $N); do
  for lib in "$@"; do
    if [ "$lib" = default ]; then unset MAPF_HIP_LIB; else export MAPF_HIP_LIB=$LIBDIR/$lib; fi
    for f in "${FLAGS[@]}"; do
      echo -n "[$lib] [$f] "
      python3 bench.py $f --steps 10 --warmup 3 --repeats 3 --no-side-legs --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f G  frac %.4f  %s' % (d['value']/1e9, d['roofline']['frac'], d['roofline']['kernel'][:90]))"
    done
  done
done
