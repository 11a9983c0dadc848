#!/usr/bin/env python3
"""One mapf_transitions configuration, launched a few times with the outputs reserved once -- the program rocprofv3 wraps
(tools/exp/transitions_prof.sh):   python3 tools/prof_transitions.py <agents> <queries> [reps] [compact]
room-32-32-4, random distinct query cells, random joint actions (the queries of tools/bench_transitions.py / bench.py's
`transitions` leg).  Prints one line: branches, HIP-event ms per launch, branches/s, bytes written/s."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'gym-mapf_amd'), ROOT]
import bench  # noqa: E402


if __name__ == '__main__':
    A, N = int(sys.argv[1]), int(sys.argv[2])
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    compact = len(sys.argv) > 4 and sys.argv[4] == 'compact'
    r = bench.transitions_rate(A, N, reps=reps, blocks=2, compact=compact)
    print(r, flush=True)
