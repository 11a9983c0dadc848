#!/usr/bin/env python3
"""env.P at 8 agents x 20000 queries measures ~100 or ~130 G branches/s from one process to the next (profiles/
r05_transitions_magic_digits_ab.txt).  Inside ONE process: several sets of output arrays (all kept alive, so each set has its
own addresses), each timed in turn, three rounds.  A rate that follows the SET says placement; one that follows the ROUND says
time; one that is constant per process says something the process got at start-up.
    python3 tools/exp/transitions_modes.py [agents=8] [queries=20000] [sets=5]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, 'gym-mapf_amd'), ROOT]
import bench  # noqa: E402,F401
import torch  # noqa: E402
from gym_mapf_amd.envs import map_name_to_files  # noqa: E402
from gym_mapf_amd.envs.grid import MapfGrid  # noqa: E402
from gym_mapf_amd.envs.utils import parse_map_file, parse_scen_file  # noqa: E402
from gym_mapf_amd.envs.vec_env import OptimizationCriteria, VecMapfEnv  # noqa: E402

A = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
S = int(sys.argv[3]) if len(sys.argv) > 3 else 5
grid = MapfGrid(parse_map_file(map_name_to_files('room-32-32-4', 6)[0]))
starts, goals = parse_scen_file(map_name_to_files('room-32-32-4', 6)[1], A)
env = VecMapfEnv(grid, A, starts, goals, 0.2, -1000.0, 100.0, -1.0, OptimizationCriteria.Makespan, n_envs=1,
                 device=torch.cuda.current_device(), device_arrays=True)
rs = np.random.RandomState(0)
V = env.n_cells
local = np.stack([rs.choice(V, A, replace=False) for _ in range(N)]).astype(np.uint16)
acts = rs.randint(0, 5, size=(N, A)).astype(np.uint8)
lt = torch.from_numpy(local.view(np.int16)).cuda().view(torch.uint16)
at = torch.from_numpy(acts).cuda()
sets = [env.transitions_compact(lt, at) for _ in range(S)]
env.sync()
branches = int(sets[0]['count'].to(torch.int64).sum().item())
t_end = time.perf_counter() + 0.2
while time.perf_counter() < t_end:
    env.transitions_compact(lt, at, out=sets[0])
    env.sync()
for rnd in range(3):
    rates = []
    for res in sets:
        env.sync()
        env.timer_begin()
        for _ in range(10):
            env.transitions_compact(lt, at, out=res)
        rates.append(branches / (env.timer_end() / 10 * 1e-3) / 1e9)
    print('round %d: ' % rnd + '  '.join('%6.1f' % r for r in rates) + '   G branches/s per output set; next-array addresses mod 2 MiB: ' +
          ' '.join('%x' % (res['next'].data_ptr() % (2 << 20)) for res in sets), flush=True)
env.close()
