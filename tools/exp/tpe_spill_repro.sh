#!/bin/bash
# Builds lib/variants/libmapf_tpe16.so: the thread-per-env rollout kernels for A = 7..16 (which need SGPR spills) are
# dispatched; tools/exp/tpe_spill_repro.py then runs them against the C oracle (round-1 note: wrong lanes seen once).
set -e
cd "$(dirname "$0")/../../gym-mapf_amd/csrc"
B=build
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I../../include -Wall -DMAPF_TPE_ROLLOUT_MAX=16"
for g in 0 1 2 3; do hipcc $F -DMAPF_GROUP=$g -c mapf_kernels.hip -o $B/tpe16_g$g.o & done
hipcc $F -c mapf_capi.hip -o $B/tpe16_capi.o &
wait
mkdir -p ../gym_mapf_amd/lib/variants
hipcc --offload-arch=gfx950 -shared -fPIC -o ../gym_mapf_amd/lib/variants/libmapf_tpe16.so $B/tpe16_capi.o $B/mapf_dispatch.o $B/mapf_lg_kernels.o \
  $B/mapf_lg_rollout.o $B/mapf_lq_rollout_k4_r1.o $B/mapf_lq_rollout_k4_r0.o $B/mapf_lq_rollout_k2_r1.o $B/mapf_lq_rollout_k2_r0.o $B/mapf_transitions.o \
  $B/tpe16_g0.o $B/tpe16_g1.o $B/tpe16_g2.o $B/tpe16_g3.o
echo built libmapf_tpe16.so
