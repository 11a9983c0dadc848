#!/usr/bin/env python3
"""Throughput of mapf_transitions (env.P[s][a] enumeration, reference mapf_env.py:448-478): every (state, action)
pair of a 2-agent empty-8-8 env (what a value-iteration sweep asks for) and random queries of 4 / 8 agents on
room-32-32-4, device arrays.  Informational."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'gym-mapf_amd'), ROOT]
import torch  # noqa: E402
from gym_mapf_amd.envs import map_name_to_files  # noqa: E402
from gym_mapf_amd.envs.grid import MapfGrid  # noqa: E402
from gym_mapf_amd.envs.utils import parse_map_file, parse_scen_file  # noqa: E402
from gym_mapf_amd.envs.vec_env import OptimizationCriteria, VecMapfEnv  # noqa: E402


def run(name, map_name, scen, A, local, acts, M):
    grid = MapfGrid(parse_map_file(map_name_to_files(map_name, scen)[0]))
    starts, goals = parse_scen_file(map_name_to_files(map_name, scen)[1], A)
    env = VecMapfEnv(grid, A, starts, goals, 0.2, -1000.0, 100.0, -1.0, OptimizationCriteria.Makespan, n_envs=1,
                     device_arrays=True)
    lt = torch.from_numpy(local.view(np.int16)).cuda().view(torch.uint16)
    at = torch.from_numpy(acts).cuda()
    res = env.transitions(lt, at, max_branches=M)
    env.sync()
    reps = 5
    env.timer_begin()
    for _ in range(reps):
        env.transitions(lt, at, max_branches=M, out=res)          # the output arrays are reserved once
    ms = env.timer_end() / reps
    branches = int(res['count'].to(torch.int64).sum().item())
    # the same calls letting every call allocate its own output arrays (what profiles/r02_transitions.txt timed): the
    # HIP-event interval then contains torch's allocator -- hipMalloc of N * M * (2A + 18) bytes once the cache is cold
    env.sync()
    torch.cuda.empty_cache()                                      # cold allocator, as in a fresh process
    env.timer_begin()
    for _ in range(reps):
        fresh = env.transitions(lt, at, max_branches=M)           # (the previous `fresh` is still alive while this one is allocated)
    ms_alloc = env.timer_end() / reps
    del fresh
    reserved = local.shape[0] * M * (2 * A + 18)
    print('%-44s queries %8d  branches %9d  reserved %7.1f MB  %.3f ms  -> %6.1f M queries/s, %7.1f M branches/s, %6.1f GB/s written'
          '   | allocating per call: %.3f ms' % (
              name, local.shape[0], branches, reserved / 1e6, ms, local.shape[0] / ms / 1e3, branches / ms / 1e3,
              branches * (2 * A + 18) / ms / 1e6, ms_alloc), flush=True)
    env.close()
    del res
    torch.cuda.empty_cache()


if __name__ == '__main__':
    V = 64
    s0, s1, a0, a1 = np.meshgrid(np.arange(V), np.arange(V), np.arange(5), np.arange(5), indexing='ij')
    local = np.stack([s0.ravel(), s1.ravel()], 1).astype(np.uint16)
    acts = np.stack([a0.ravel(), a1.ravel()], 1).astype(np.uint8)
    run('empty-8-8, 2 agents: all (s, a) pairs', 'empty-8-8', 1, 2, local, acts, 9)
    run('empty-8-8, 2 agents: all (s, a) pairs x 32', 'empty-8-8', 1, 2, np.tile(local, (32, 1)), np.tile(acts, (32, 1)), 9)
    rs = np.random.RandomState(0)
    for A, M, N in ((4, 81, 200000), (4, 81, 2000000), (8, 6561, 2000), (8, 6561, 20000)):
        local = np.stack([rs.choice(682, A, replace=False) for _ in range(N)]).astype(np.uint16)
        acts = rs.randint(0, 5, size=(N, A)).astype(np.uint8)
        run('room-32-32-4, %d agents: random queries' % A, 'room-32-32-4', 6, A, local, acts, M)
