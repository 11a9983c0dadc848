#!/bin/bash
out=gpurun_out/b1024.txt; : > $out
export MAPF_HIP_LIB=$PWD/gym-mapf_amd/gym_mapf_amd/lib/variants/libmapf_b1024.so
for rep in 1 2; do
  for big in "" 1 2; do
    if [ -z "$big" ]; then unset MAPF_EXP_BLOCK1024; else export MAPF_EXP_BLOCK1024=$big; fi
    echo "== MAPF_EXP_BLOCK1024=$big" >> $out
    timeout -k 10 120 python tools/exp/rates.py c3 c3x2 c4 >> $out 2>&1 || exit 1
  done
done
grep -v amdgpu.ids $out | cut -c1-150
