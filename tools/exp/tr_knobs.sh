#!/bin/bash
for rows in 4096 2048 1024 512; do
for a in "8 20000" "8 2000" "7 20000" "8 200000" "8 50000"; do
  echo "== rows $rows, $a: $(MAPF_TR_ROWS=$rows python3 tools/prof_transitions.py $a 10 compact 2>&1 | grep -o "'ms_per_launch_hip_events': [0-9.]*\|'frac': [0-9.]*" | tr '\n' ' ')"
done; done
