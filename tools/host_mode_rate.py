#!/usr/bin/env python3
"""PCIe-inclusive rates: the same workload as bench.py but through HOST (numpy) arrays, i.e. every call copies its
inputs to the device and its outputs back (for DESIGN.md; never the bench `value`)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'gym-mapf_amd'), ROOT]
import bench  # noqa: E402
from gym_mapf_amd.envs.vec_env import OptimizationCriteria, VecMapfEnv  # noqa: E402

E, A, T = 65536, 8, 64
grid, _, nbr, start, goal = bench.workload_tables(bench.CONFIGS['c3'], E, 0)
env = VecMapfEnv(grid, A, None, None, 0.2, -1000.0, 100.0, -1.0, OptimizationCriteria.Makespan, seed=42,
                 start_local=start, goal_local=goal)
acts = env.fill_random_actions(0, T)
for k in range(4):
    env.step(acts[k], auto_reset=True)
t0 = time.perf_counter()
n = 200
for k in range(n):
    env.step(acts[k % T], auto_reset=True)
dt = time.perf_counter() - t0
print('host-mode mapf_step   : %.1f us/step, %.2f G agent-steps/s (H2D actions + D2H all outputs per step)' % (dt / n * 1e6, n * E * A / dt / 1e9))
env.rollout(T, actions=acts, auto_reset=True, record=True)
t0 = time.perf_counter()
n = 10
for _ in range(n):
    env.rollout(T, actions=acts, auto_reset=True, record=True)
dt = time.perf_counter() - t0
print('host-mode mapf_rollout: %.1f us/step, %.2f G agent-steps/s (T=64, recorded trajectory copied back)' % (dt / n / T * 1e6, n * T * E * A / dt / 1e9))
t0 = time.perf_counter()
for _ in range(n):
    env.rollout(T, actions=None, auto_reset=True, record=False)
dt = time.perf_counter() - t0
print('host-mode mapf_rollout: %.1f us/step, %.2f G agent-steps/s (T=64, in-kernel policy, returns only)' % (dt / n / T * 1e6, n * T * E * A / dt / 1e9))
