#!/usr/bin/env python3
"""One mapf_step launch per env-step (every output written, device arrays): time per launch and fraction of the HBM
roofline (5 + 18/A algorithmic bytes per agent-step) as the batch grows -- where does the launch-latency-bound single
step cross a given fraction?      python tools/single_step_scaling.py [n_agents]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'gym-mapf_amd'), ROOT]
import numpy as np
import torch
import bench
from gym_mapf_amd.envs.vec_env import OptimizationCriteria, VecMapfEnv

cfg = bench.CONFIGS['c3']
A = cfg['agents']
torch.cuda.set_device(0)
print('room-32-32-4, %d agents, slip 0.2, auto-reset, uniform-random actions resident in HBM' % A)
for E in (4096, 16384, 65536, 131072, 262144, 524288, 1048576, 2097152):
    grid, _, nbr, start, goal = bench.workload_tables(cfg, E, 0)
    env = VecMapfEnv(grid, A, None, None, cfg['fail_prob'], -1000.0, 100.0, -1.0, OptimizationCriteria.Makespan, seed=42,
                     device_arrays=True, start_local=start, goal_local=goal)
    ring = 16
    actions = env.fill_random_actions(0, ring)
    out, calls = None, []
    for r in range(ring):
        call, out = env.prepare_step(actions[r], auto_reset=True, out=out)
        calls.append(call)
    t_end = time.perf_counter() + 0.1
    k = 0
    while time.perf_counter() < t_end:
        for _ in range(64):
            calls[k % ring](); k += 1
        env.sync()
    n = 2000 if E <= 262144 else 500
    env.timer_begin()
    for k in range(n):
        calls[k % ring]()
    us = env.timer_end() * 1e3 / n
    byts = E * A * (5 + 18.0 / A)
    print('E=%8d  %8.2f us per launch  %8.1f G agent-steps/s  %7.1f GB/s algorithmic = %.3f of 8 TB/s   %s'
          % (E, us, E * A / us / 1e3, byts / us / 1e3, byts / us / 1e3 / 8000.0, env.last_kernel('step')), flush=True)
    env.close()
    del actions, out
    torch.cuda.empty_cache()
