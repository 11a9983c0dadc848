#!/usr/bin/env python3
"""Per-launch durations of back-to-back fused rollout launches right after an idle period (clock ramp / DVFS check).
usage: python tools/exp/launch_series.py [n_launches] [envs]"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, 'gym-mapf_amd'), ROOT]
import numpy as np
import torch
import bench
from gym_mapf_amd import _native as nat
from gym_mapf_amd.envs.vec_env import OptimizationCriteria, VecMapfEnv

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
E = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
cfg = bench.CONFIGS['c3']
A, T = cfg['agents'], 256
torch.cuda.set_device(0)
grid, _, nbr, start, goal = bench.workload_tables(cfg, E, 0)
env = VecMapfEnv(grid, A, None, None, cfg['fail_prob'], -1000.0, 100.0, -1.0, OptimizationCriteria.Makespan, seed=42,
                 device_arrays=True, start_local=start, goal_local=goal)
actions = env.fill_random_actions(0, T)
res = env.rollout(T, actions=actions, auto_reset=True, record=True)
env.sync()
time.sleep(2.0)                                  # let the clocks fall back
stream = torch.cuda.ExternalStream(env.stream)
evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
io = nat.MapfRolloutIO(struct_size=ctypes.sizeof(nat.MapfRolloutIO), n_steps=T, step_flags=nat.MAPF_STEP_AUTO_RESET, accumulate=0,
                       actions=actions.data_ptr(), out_returns=res['returns'].data_ptr(), out_episodes=res['episodes'].data_ptr(),
                       out_collisions=res['collisions'].data_ptr(), rec_local=res['local'].data_ptr(), rec_reward=res['reward'].data_ptr(),
                       rec_done=res['done'].data_ptr(), rec_collision=res['collision'].data_ptr(), rec_prob=res['prob'].data_ptr())
with torch.cuda.stream(stream):
    evs[0].record(stream)
    for k in range(n):
        nat.check(env._lib.mapf_rollout(env._h, ctypes.byref(io)))
        evs[k + 1].record(stream)
env.sync()
torch.cuda.synchronize()
d = np.array([evs[k].elapsed_time(evs[k + 1]) for k in range(n)]) * 1e3
print('kernel', env.last_kernel('rollout'))
print('us per launch, launches 0..:', ' '.join('%.0f' % x for x in d[:40]))
for lo in range(0, n, 25):
    print('launches %3d-%3d: mean %.1f us  -> %.1f G agent-steps/s' % (lo, min(lo + 25, n) - 1, d[lo:lo + 25].mean(), T * E * A / d[lo:lo + 25].mean() / 1e3))
