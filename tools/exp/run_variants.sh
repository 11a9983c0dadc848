#!/bin/bash
# usage: run_variants.sh "<cases>" name1 name2 ...   (base = the in-tree library)
cases=$1; shift
out=gpurun_out/variants.txt; : > $out
for rep in 1 2; do
  for v in base "$@"; do
    if [ $v = base ]; then unset MAPF_HIP_LIB; else export MAPF_HIP_LIB=$PWD/gym-mapf_amd/gym_mapf_amd/lib/variants/libmapf_$v.so; fi
    echo "== $v" >> $out
    timeout -k 10 120 python tools/exp/rates.py $cases >> $out 2>&1 || exit 1
  done
done
cat $out
