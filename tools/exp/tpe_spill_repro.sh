#!/bin/bash
# Builds lib/variants/libmapf_tpe16.so (make target `tpe16`): the thread-per-env rollout kernels for A = 7..16, which need
# SGPR spills and are not part of the shipped library, compiled in and dispatched; tools/exp/tpe_spill_repro.py then runs
# them against the C oracle (round-1 note: wrong lanes seen once, never reproduced -- profiles/r02_tpe_spill_repro.txt).
set -e
make -j8 -C "$(dirname "$0")/../../gym-mapf_amd/csrc" tpe16
echo built libmapf_tpe16.so
