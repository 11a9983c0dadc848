#!/bin/bash
# experiment: transitions_rows_kernel launch-shape knobs (temporary env overrides), compacted rows
for a in "8 20000" "8 2000" "7 20000" "6 100000" "5 200000" "4 2000000" "4 200000" "3 2000000" "2 1000000"; do
for cfg in "" "MAPF_TR_WPB=4" "MAPF_TR_WPB=1" "MAPF_TR_WPB=2"; do
  echo "== $a $cfg: $(env $cfg python3 tools/prof_transitions.py $a 10 compact 2>&1 | grep -o "'ms_per_launch_hip_events': [0-9.]*\|'frac': [0-9.]*" | tr '\n' ' ')"
done; done
