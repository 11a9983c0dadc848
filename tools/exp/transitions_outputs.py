#!/usr/bin/env python3
"""Which of env.P's five output streams costs what: mapf_transitions_compact with some output arrays left out (null pointers:
the kernel's ALL_OUT = false instance skips those stores), 8 agents x 20000 queries, ms per launch and GB/s of what IS written.
    python3 tools/exp/transitions_outputs.py [agents=8] [queries=20000]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, 'gym-mapf_amd'), ROOT]
import bench  # noqa: E402,F401
import torch  # noqa: E402
from gym_mapf_amd import _native as nat  # noqa: E402
from gym_mapf_amd.envs import map_name_to_files  # noqa: E402
from gym_mapf_amd.envs.grid import MapfGrid  # noqa: E402
from gym_mapf_amd.envs.utils import parse_map_file, parse_scen_file  # noqa: E402
from gym_mapf_amd.envs.vec_env import OptimizationCriteria, VecMapfEnv  # noqa: E402

A = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
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
res = env.transitions_compact(lt, at)
env.sync()
branches = int(res['count'].to(torch.int64).sum().item())
R, M = int(res['prob'].shape[0]), 3 ** A
sizes = {'next': 2 * A, 'prob': 8, 'reward': 8, 'done': 1, 'collision': 1}


def launch(keep):
    p = lambda k, dt, shape: env._ptr(res[k], dt, shape, k) if k in keep else None  # noqa: E731
    nat.check(env._lib.mapf_transitions_compact(
        env._h, N, env._ptr(lt, np.uint16, (N, A), 'local'), env._ptr(at, np.uint8, (N, A), 'actions'), None, 0, M, R,
        env._ptr(res['offset'], np.uint64, (N + 1,), 'offset'), env._ptr(res['count'], np.uint32, (N,), 'count'),
        p('next', np.uint16, (R, A)), p('prob', np.float64, (R,)), p('reward', np.float64, (R,)), p('done', np.uint8, (R,)),
        p('collision', np.uint8, (R,))))


t_end = time.perf_counter() + 0.2
while time.perf_counter() < t_end:
    launch(set(sizes))
    env.sync()
for keep in (set(sizes), {'next', 'prob', 'reward'}, {'next'}, {'prob', 'reward'}, {'done', 'collision'}, {'prob'}, set(), set(sizes)):
    ms = []
    for _ in range(3):
        env.sync()
        env.timer_begin()
        for _ in range(10):
            launch(keep)
        ms.append(env.timer_end() / 10)
    m = sorted(ms)[1]
    nbytes = branches * sum(sizes[k] for k in keep)
    print('%-44s %.3f ms  %7.1f GB/s written  (%s)' % ('+'.join(sorted(keep)) or 'nothing written', m, nbytes / (m * 1e-3) / 1e9, env.last_kernel('transitions')[:40]), flush=True)
env.close()
