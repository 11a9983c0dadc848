#!/usr/bin/env python3
"""Where a packed single step (lq_step_kernel) spends its time: per-wave stage stamps from the diagnostic build
(`make -C gym-mapf_amd/csrc step_stamps` -> lib/variants/libmapf_hip_step_stamps.so; never shipped).

    MAPF_HIP_LIB=.../lib/variants/libmapf_hip_step_stamps.so python tools/step_stamps.py [n_envs] [graph|plain] [c3|c4|c5]

Every wave records s_memrealtime (100 MHz, chip-wide) at entry and exit and s_memtime (shader cycles) at the stages
in between, each stage stamp behind a full wait for the memory operations issued so far.  The waits make this build a
little slower than the shipped kernel; what is read off is the SHAPE: how long each trip takes, how far the waves'
starts are spread, how much of a launch is the tail."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'gym-mapf_amd'), ROOT]
import torch  # noqa: E402
import bench  # noqa: E402
from gym_mapf_amd.envs.vec_env import OptimizationCriteria, VecMapfEnv  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
mode = sys.argv[2] if len(sys.argv) > 2 else 'graph'
cfg = bench.CONFIGS[sys.argv[3] if len(sys.argv) > 3 else 'c3']
A = cfg['agents']
grid, _, nbr, start, goal = bench.workload_tables(cfg, E, 0)
env = VecMapfEnv(grid, A, None, None, cfg['fail_prob'], -1000.0, 100.0, -1.0, OptimizationCriteria.Makespan, seed=42,
                 device_arrays=True, start_local=start, goal_local=goal)
N = 32
actions = env.fill_random_actions(0, N)
stamps = torch.zeros((E, A), dtype=torch.float64, device='cuda')     # travels as `uniforms`: 12 x u64 per wave
out = None
calls = []
if mode == 'graph':
    env.graph_begin()
for k in range(N):
    call, out = env.prepare_step(actions[k], uniforms=stamps, auto_reset=True, out=out, write_local=False)
    calls.append(call)
    if mode == 'graph':
        call()
if mode == 'graph':
    graph = env.graph_end()
    for _ in range(20):
        graph.launch(1)
else:
    for _ in range(20):
        for call in calls:
            call()
env.sync()
print('kernel:', env.last_kernel('step'), '| mode', mode, '| envs', E)
waves = E * (A // 4) // 64
raw = stamps.view(torch.int64).cpu().numpy().reshape(-1)[:waves * 12].reshape(waves, 12)
real0, real1 = raw[:, 0], raw[:, 1]
t_first = real0.min()
names = ['argument block arrived', 'six Philox rounds done', 'first loads arrived (state, actions, scen)',
         'gathers issued + four rounds done', 'gathers arrived (table rows, scen rows)', 'everything computed',
         'stores issued', 'stores acknowledged']
cyc = raw[:, 2:10].astype(np.float64)
print('last launch of the run: %d waves; chip-wide clock (100 MHz ticks -> us):' % waves)
print('  wave entry  after the first wave: median %.2f us, p90 %.2f, max %.2f' % tuple(
    np.percentile((real0 - t_first) / 100.0, [50, 90, 100])))
print('  wave exit   after the first wave: median %.2f us, p90 %.2f, max %.2f  (= the launch as the waves see it)' % tuple(
    np.percentile((real1 - t_first) / 100.0, [50, 90, 100])))
life = (real1 - real0) / 100.0
print('  wave lifetime: median %.2f us, p10 %.2f, p90 %.2f, max %.2f' % tuple(np.percentile(life, [50, 10, 90, 100])))
mhz = np.median(cyc[:, 7] / np.maximum(life, 1e-9))
print('  shader clock seen by s_memtime: %.0f MHz (stage cycles / lifetime)' % mhz)
print('stage stamps, shader cycles since wave entry (median / p10 / p90) and the median step between stages in us:')
prev = np.zeros(waves)
for i, n in enumerate(names):
    med, p10, p90 = np.percentile(cyc[:, i], [50, 10, 90])
    print('  %-44s %7.0f / %7.0f / %7.0f   +%.2f us' % (n, med, p10, p90, np.median(cyc[:, i] - prev) / mhz))
    prev = cyc[:, i]
xcc = raw[:, 11] & 0xF
print('waves per XCC:', np.bincount(xcc.astype(np.int64), minlength=8).tolist())
for xid in range(8):
    sel = xcc == xid
    if sel.any():
        print('  XCC %d: entry median %.2f us, exit median %.2f us' % (xid, np.median((real0[sel] - t_first) / 100.0), np.median((real1[sel] - t_first) / 100.0)))
env.close()
