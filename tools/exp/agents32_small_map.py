#!/usr/bin/env python3
"""32 agents on a map whose FULL move table fits the LDS (room-32-32-4, random distinct start / goal cells): the fused
rollout under default dispatch (occupancy bitmaps behind the full table) against the all-pairs forms (MAPF_TUNE bitmap_pairs=0 /
MAPF_TUNE k=8), at two batch sizes.
    gpurun -- 'python tools/exp/agents32_small_map.py'"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    sys.path[:0] = [os.path.join(ROOT, 'gym-mapf_amd'), os.path.join(ROOT, 'tools'), ROOT]
    import bench_configs as bc
    E = int(sys.argv[1])
    # (no room-32-32-4 scenario constructs with 32 agents -- some of the first 32 sit on obstacle cells under the reference's
    # row / column convention -- so: seeded random distinct start / goal cells, as bench.py draws them for configs[4])
    import numpy as np
    import bench
    from gym_mapf_amd.envs import map_name_to_files
    from gym_mapf_amd.envs.grid import MapfGrid
    from gym_mapf_amd.envs.utils import parse_map_file
    g = MapfGrid(parse_map_file(map_name_to_files('room-32-32-4', 1)[0]))
    V = len(g.tables()[0])
    s = bench.random_distinct_cells(V, 32, np.arange(E), 1)
    t = bench.random_distinct_cells(V, 32, np.arange(E), 2)
    bc.measure('room-32-32-4, 32 agents [%s]' % os.environ.get('FORM', 'default'), g, s, t, 32, 0.2)
else:
    for E in ('16384', '65536'):
        for form in ('', 'MAPF_TUNE=bitmap_pairs=0', 'MAPF_TUNE=k=8'):
            env = dict(os.environ, FORM=form or 'default')
            if form:
                k, v = form.split('=', 1)
                env[k] = v
            subprocess.run([sys.executable, os.path.abspath(__file__), E], env=env, check=False)
