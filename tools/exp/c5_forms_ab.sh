#!/bin/bash
# A/B of the 32-agent rollout forms on ONE box (C5 whole, 131072 envs, and C5's share, 16384): the default dispatch against
# the forms the environment can still pin -- MAPF_TUNE=k=8 (eight agents per lane, all agent pairs), MAPF_TUNE=bitmap_block=512 /
# 1024 (occupancy bitmaps, 64 / 128 envs per block), MAPF_TUNE=bitmap_staycol=0 (bitmaps behind the four-column table),
# MAPF_TUNE=bitmap_pairs=0 (four per lane, all pairs).
#   gpurun -- 'bash tools/exp/c5_forms_ab.sh 2'            # every form
#   gpurun -- 'bash tools/exp/c5_forms_ab.sh 2 default MAPF_TUNE=bitmap_staycol=0'
N=${1:-2}; shift || true
forms=("$@")
[ ${#forms[@]} -eq 0 ] && forms=(default MAPF_TUNE=k=8 MAPF_TUNE=bitmap_block=512 MAPF_TUNE=bitmap_block=1024 MAPF_TUNE=bitmap_staycol=0 MAPF_TUNE=bitmap_pairs=0)
for i in $(seq $N); do
  for cfgflags in "--config c5" "--config c5 --envs 16384"; do
    for form in "${forms[@]}"; do
      echo -n "[$form] [$cfgflags] "
      setting=$form; [ "$form" = default ] && setting=MAPF_TUNE=
      env $setting python3 bench.py $cfgflags --steps 10 --warmup 3 --repeats 3 --no-side-legs --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f G  frac %.4f  %s' % (d['value']/1e9, d['roofline']['frac'], d['roofline']['kernel'][:105]))"
    done
  done
done
