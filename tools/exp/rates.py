#!/usr/bin/env python3
"""Steady-state fused-rollout rate (recorded trajectory, streamed actions) for a few batches; one library per process.
usage: MAPF_HIP_LIB=... python tools/exp/rates.py [case ...]     cases: c3 c4s c4s_quad c4s_pair c5s c2 c3x2"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, 'gym-mapf_amd'), ROOT]
import numpy as np
import torch
import bench
from gym_mapf_amd import _native as nat
from gym_mapf_amd.envs.vec_env import OptimizationCriteria, VecMapfEnv

CASES = {'c3': ('c3', 65536, {}), 'c3x2': ('c3', 131072, {}), 'c4s': ('c4', 32768, {}),
         'c4s_quad': ('c4', 32768, {'MAPF_TUNE': 'quad_min_lanes=0'}), 'c4s_pair': ('c4', 32768, {'MAPF_TUNE': 'quad_lanes=0'}),
         'c3_pair': ('c3', 65536, {'MAPF_TUNE': 'quad_lanes=0'}), 'c4s_k2': ('c4', 32768, {'MAPF_TUNE': 'k=2'}), 'c3_k2': ('c3', 65536, {'MAPF_TUNE': 'k=2'}),
         'c5s_k2': ('c5', 16384, {'MAPF_TUNE': 'k=2'}), 'c3_k8': ('c3', 65536, {'MAPF_TUNE': 'k=8'}), 'c3x2_k4': ('c3', 131072, {'MAPF_TUNE': 'k=4'}),
         'c4_k4': ('c4', 262144, {'MAPF_TUNE': 'k=4'}), 'c5_k4': ('c5', 131072, {'MAPF_TUNE': 'k=4'}), 'c5s_k8': ('c5', 16384, {'MAPF_TUNE': 'k=8'}),
         'c5s': ('c5', 16384, {}), 'c5': ('c5', 131072, {}), 'c2': ('c2', 4096, {}), 'c4': ('c4', 262144, {})}
torch.cuda.set_device(0)
T = 256
for case in (sys.argv[1:] or ['c3', 'c4s', 'c5s']):
    name, E, envvars = CASES[case]
    os.environ.pop('MAPF_TUNE', None)
    os.environ.update(envvars)
    cfg = bench.CONFIGS[name]
    A = cfg['agents']
    grid, _, nbr, start, goal = bench.workload_tables(cfg, E, 0)
    env = VecMapfEnv(grid, A, None, None, cfg['fail_prob'], -1000.0, 100.0, -1.0, OptimizationCriteria.Makespan, seed=42,
                     device_arrays=True, start_local=start, goal_local=goal)
    actions = env.fill_random_actions(0, T)
    res = env.rollout(T, actions=actions, auto_reset=True, record=True)
    io = nat.MapfRolloutIO(struct_size=ctypes.sizeof(nat.MapfRolloutIO), n_steps=T, step_flags=nat.MAPF_STEP_AUTO_RESET, accumulate=0,
                           actions=actions.data_ptr(), out_returns=res['returns'].data_ptr(), out_episodes=res['episodes'].data_ptr(),
                           out_collisions=res['collisions'].data_ptr(), rec_local=res['local'].data_ptr(), rec_reward=res['reward'].data_ptr(),
                           rec_done=res['done'].data_ptr(), rec_collision=res['collision'].data_ptr(), rec_prob=res['prob'].data_ptr())
    t_end = time.perf_counter() + 0.15                      # pre-roll: let the clocks settle
    while time.perf_counter() < t_end:
        for _ in range(10):
            nat.check(env._lib.mapf_rollout(env._h, ctypes.byref(io)))
        env.sync()
    best = 1e9
    for rep in range(3):
        n = 40
        env.timer_begin()
        for _ in range(n):
            nat.check(env._lib.mapf_rollout(env._h, ctypes.byref(io)))
        ms = env.timer_end() / n
        best = min(best, ms)
    rate = T * E * A / (best * 1e-3)
    print('%-9s E=%6d A=%2d %8.1f us/launch %7.1f G agent-steps/s  frac %.3f  %s' % (
        case, E, A, best * 1e3, rate / 1e9, rate * (5 + 18.0 / A) / 8e12, env.last_kernel('rollout')), flush=True)
    env.close()
    del res, actions
    torch.cuda.empty_cache()
