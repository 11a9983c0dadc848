#!/usr/bin/env python3
"""What a SHORT fused rollout costs: T in {8, 16, 32, 64, 128, 256} env-steps per mapf_rollout launch on
the bench batch (configs[2], 65536 envs x 8 agents) and configs[4]'s share of one GPU (16384 envs x 32 agents), recorded
trajectory, streamed actions, three ways:
  (a) the C ABI with everything preallocated (bench.py's own path: one ctypes call per launch);
  (b) VecMapfEnv.rollout(record=True, out=previous result)  -- the wrapper, buffers reused;
  (c) VecMapfEnv.rollout(record=True)                        -- the wrapper allocating its eight arrays per call;
  (d) the launches of (a) recorded into a hipGraph (16 per replay)  -- no host enqueue between the kernels.
HIP-event time per launch (median of 5 blocks of 20 launches); a least-squares line t(T) = fixed + T * per_step through (a)
separates the kernel's fixed cost per launch (table staging, first loads, the drain of the last step) from its step rate.

    python tools/rollout_T_sweep.py > profiles/r05_rollout_T_sweep.txt       (T_SWEEP_POLICY=1 ... c3 65536: the in-kernel policy)
"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'gym-mapf_amd'), ROOT]
import torch  # noqa: E402
import bench  # noqa: E402
from gym_mapf_amd import _native as nat  # noqa: E402
from gym_mapf_amd.envs.vec_env import OptimizationCriteria, VecMapfEnv  # noqa: E402

TS = tuple(int(x) for x in os.environ['T_SWEEP'].split(',')) if os.environ.get('T_SWEEP') else (8, 16, 32, 64, 128, 256)
POLICY = os.environ.get('T_SWEEP_POLICY') == '1'   # actions = NULL: the in-kernel policy instead of streamed actions


def timed(env, fn, n=20, blocks=5):
    for _ in range(6):
        fn()
    env.sync()
    ms = []
    for _ in range(blocks):
        env.sync()
        env.timer_begin()
        for _ in range(n):
            fn()
        ms.append(env.timer_end() / n)
        env.sync()
    return sorted(ms)[len(ms) // 2]


def sweep(name, n_envs):
    cfg = bench.CONFIGS[name]
    A, E = cfg['agents'], n_envs
    grid, _, nbr, start, goal = bench.workload_tables(cfg, E, 0)
    env = VecMapfEnv(grid, A, None, None, cfg['fail_prob'], -1000.0, 100.0, -1.0, OptimizationCriteria.Makespan, seed=42,
                     device_arrays=True, start_local=start, goal_local=goal)
    Tmax = max(TS)
    actions = env.fill_random_actions(0, Tmax)
    rec = {'local': env._empty((Tmax, E, A), np.uint16), 'reward': env._empty((Tmax, E), np.float64), 'prob': env._empty((Tmax, E), np.float64),
           'done': env._empty((Tmax, E), np.uint8), 'collision': env._empty((Tmax, E), np.uint8)}
    acc = {'returns': torch.zeros(E, dtype=torch.float64, device='cuda'),
           'episodes': torch.zeros(E, dtype=torch.int32, device='cuda').view(torch.uint32),
           'collisions': torch.zeros(E, dtype=torch.int32, device='cuda').view(torch.uint32)}
    print('%s: %s map, %d agents, %d envs, recorded trajectory, %s' % (cfg['baseline'], cfg['map'], A, E, 'in-kernel policy' if POLICY else 'streamed actions'))
    print('    T   | (a) C ABI, preallocated      | (b) rollout(out=...)         | (c) rollout() allocating     | (d) hipGraph replay of (a)   | kernel')
    rows, rows_d = [], []
    for T in TS:
        io = nat.MapfRolloutIO(struct_size=ctypes.sizeof(nat.MapfRolloutIO), n_steps=T, step_flags=nat.MAPF_STEP_AUTO_RESET, accumulate=1,
                               actions=None if POLICY else actions.data_ptr(), out_returns=acc['returns'].data_ptr(), out_episodes=acc['episodes'].data_ptr(),
                               out_collisions=acc['collisions'].data_ptr(), rec_local=rec['local'].data_ptr(), rec_reward=rec['reward'].data_ptr(),
                               rec_done=rec['done'].data_ptr(), rec_collision=rec['collision'].data_ptr(), rec_prob=rec['prob'].data_ptr())
        a = timed(env, lambda: nat.check(env._lib.mapf_rollout(env._h, ctypes.byref(io))))
        # (d) the same launch recorded into a hipGraph, 16 per replay: the host is out of the loop, what is left is the kernel
        env.graph_begin()
        for _ in range(16):
            nat.check(env._lib.mapf_rollout(env._h, ctypes.byref(io)))
        graph = env.graph_end()
        d = timed(env, lambda: graph.launch(1), n=4) / 16.0
        graph.close()
        acts = None if POLICY else actions[:T]
        res = env.rollout(T, actions=acts, auto_reset=True, record=True)
        b = timed(env, lambda: env.rollout(T, actions=acts, auto_reset=True, record=True, out=res))
        c = timed(env, lambda: env.rollout(T, actions=acts, auto_reset=True, record=True))
        del res
        rate = lambda ms: T * E * A / (ms * 1e-3) / 1e9
        frac = lambda ms: T * E * A * bench.bytes_per_agent_step(A) / (ms * 1e-3) / 1e9 / bench.HBM_PEAK_GBS
        rows.append((T, a))
        rows_d.append((T, d))
        print('  %4d   | %8.2f us %6.1f G  %.3f | %8.2f us %6.1f G  %.3f | %8.2f us %6.1f G  %.3f | %8.2f us %6.1f G  %.3f | %s' % (
            T, a * 1e3, rate(a), frac(a), b * 1e3, rate(b), frac(b), c * 1e3, rate(c), frac(c), d * 1e3, rate(d), frac(d),
            env.last_kernel('rollout')[:44]), flush=True)
    t = np.array([r[0] for r in rows], float)
    y = np.array([r[1] for r in rows], float) * 1e3
    per_step, fixed = np.polyfit(t, y, 1)
    print('  (a) as a line: %.2f us fixed per launch + %.3f us per env-step  ->  the fixed part is %.0f %% of a T = 32 launch, %.0f %% at T = 256'
          % (fixed, per_step, 100 * fixed / (fixed + 32 * per_step), 100 * fixed / (fixed + 256 * per_step)))
    t = np.array([r[0] for r in rows_d], float)
    y = np.array([r[1] for r in rows_d], float) * 1e3
    per_step, fixed = np.polyfit(t, y, 1)
    print('  (d) as a line: %.2f us fixed per launch + %.3f us per env-step  ->  %.0f %% of a T = 32 launch, %.0f %% at T = 256  (the kernel alone: no host enqueue)'
          % (fixed, per_step, 100 * fixed / (fixed + 32 * per_step), 100 * fixed / (fixed + 256 * per_step)))
    env.close()
    del rec, acc, actions
    torch.cuda.empty_cache()


if __name__ == '__main__':
    if len(sys.argv) > 2:
        sweep(sys.argv[1], int(sys.argv[2]))       # one batch: python tools/rollout_T_sweep.py c3 65536   (T_SWEEP=1,2,4 picks the lengths)
    else:
        sweep('c3', 65536)
        sweep('c5', 16384)
        sweep('c4', 32768)
