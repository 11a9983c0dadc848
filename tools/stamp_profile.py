#!/usr/bin/env python3
"""Where does a rollout step spend its cycles?  Runs the bench workload on the DIAGNOSTIC library
(`make -C gym-mapf_amd/csrc stamps`, s_memtime stamps around the loop's segments; the segment sums overwrite the
episode counters, so this build's outputs are not valid results) and prints median cycles per step and segment.

    make -C gym-mapf_amd/csrc stamps
    MAPF_HIP_LIB=gym-mapf_amd/gym_mapf_amd/lib/variants/libmapf_hip_stamps.so python tools/stamp_profile.py [envs] [c3|c5] [T=64]
    (MAPF_TUNE=k=2 profiles the packed layout with two agents per lane, MAPF_TUNE=quad_lanes=0 the lane-group kernel)

Stamps serialise the segments (a fence on each side), so read the SHARES, not the total (cdna_hip_programming.md
section 7, in-kernel stamps).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'gym-mapf_amd'), ROOT]
import bench  # noqa: E402
from gym_mapf_amd.envs.vec_env import OptimizationCriteria, VecMapfEnv  # noqa: E402

if 'stamps' not in os.environ.get('MAPF_HIP_LIB', ''):
    raise SystemExit('set MAPF_HIP_LIB to the stamps build (see the docstring)')
cfg_name = sys.argv[2] if len(sys.argv) > 2 else 'c3'
cfg = bench.CONFIGS[cfg_name]
E, A = (int(sys.argv[1]) if len(sys.argv) > 1 else 65536), cfg['agents']
T = int(sys.argv[3]) if len(sys.argv) > 3 else 64          # env-steps per launch (short launches: where do the first steps' cycles go?)
# envs per wave: 64 / (lanes per env); 8 agents: 32 with four agents per lane (default), 16 with two
_tune = dict(item.split('=', 1) for item in os.environ.get('MAPF_TUNE', '').split(',') if item)
pair_layout = _tune.get('quad_lanes') == '0' or _tune.get('k') == '2'
PER_WAVE = 64 * (2 if pair_layout else 4) // A
print('layout: %s' % ('pair (2 agents per lane)' if pair_layout else 'quad (4 agents per lane)'))
grid, _, nbr, start, goal = bench.workload_tables(cfg, E, 0)
print('config %s, envs: %d, agents: %d' % (cfg_name, E, A))
env = VecMapfEnv(grid, A, None, None, 0.2, -1000.0, 100.0, -1.0, OptimizationCriteria.Makespan, seed=42,
                 device_arrays=True, start_local=start, goal_local=goal)
actions = env.fill_random_actions(0, T)
env.sync()
if pair_layout:
    names = ['loop top (actions, delayed stores)', 'slip Philox (1 step in 4) + table read', 'sampling + probability read',
             'pair tests', 'flags + group reduce', 'probability product', 'reward / selects', 'reset handling']
else:
    names = ['loop top (actions, table read issue)', 'previous step: prob chain, totals, stores', 'slip Philox (1 step in 4)',
             'sampling (table wait) + probability read', 'pair tests', 'flags + group reduce', 'outcome request',
             'reset handling']
for acts, label in ((actions, 'streamed actions'), (None, 'in-kernel policy')):
    for record in (True,):   # (the stamps build instruments the recording kernels only)
        env.reset()
        env.rollout(T, actions=acts, auto_reset=True, record=record)
        res = env.rollout(T, actions=acts, auto_reset=True, record=record)
        env.sync()
        print('kernel:', env.last_kernel('rollout'))
        seg = res['episodes'].cpu().numpy().view(np.uint32).reshape(-1, PER_WAVE)[:, :8].astype(np.float64) / T
        med = np.median(seg, axis=0)
        print('%s, record=%s: %d cycles per wave-step' % (label, record, med.sum()))
        for n, m in zip(names, med):
            print('    %-36s %6.0f  (%4.1f %%)' % (n, m, 100 * m / med.sum()))
