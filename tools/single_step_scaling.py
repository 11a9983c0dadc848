#!/usr/bin/env python3
"""One mapf_step launch per env-step (every output written, device arrays): time per launch and fraction of the HBM
roofline (5 + 18/A algorithmic bytes per agent-step) as the batch grows -- where does the launch-latency-bound single
step cross a given fraction?  Steps are recorded into a hipGraph (16 nodes) and replayed; the next observation is the
handle's state view (out_local = NULL).  At the largest batches the action ring is also run with 4 slots instead of 16:
the working set of a step then fits the 256 MB Infinity Cache again (see the 2 M line).
      python tools/single_step_scaling.py"""
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
def run(E, ring, mode, nodes=16):
    grid, _, nbr, start, goal = bench.workload_tables(cfg, E, 0)
    env = VecMapfEnv(grid, A, None, None, cfg['fail_prob'], -1000.0, 100.0, -1.0, OptimizationCriteria.Makespan, seed=42,
                     device_arrays=True, start_local=start, goal_local=goal)
    actions = env.fill_random_actions(0, ring)
    out, calls = None, []
    if mode == 'graph':
        env.graph_begin()
    for r in range(nodes):
        call, out = env.prepare_step(actions[r % ring], auto_reset=True, out=out, write_local=False)
        calls.append(call)
        if mode == 'graph':
            call()
    if mode == 'graph':
        graph = env.graph_end()
        launch = lambda: graph.launch(1)
    else:
        def launch():
            for c in calls:
                c()
    t_end = time.perf_counter() + 0.1
    while time.perf_counter() < t_end:
        for _ in range(4):
            launch()
        env.sync()
    n = max(1, (2000 if E <= 262144 else 500) // nodes)
    env.timer_begin()
    for k in range(n):
        launch()
    us = env.timer_end() * 1e3 / (n * nodes)
    byts = E * A * (5 + 18.0 / A)
    touched = E * (A * 2 + 18 + 1 + 1) + ring * E * A     # state + outputs + scenario byte + action ring
    print('E=%8d %5s ring=%2d  %8.2f us per launch  %8.1f G agent-steps/s  %7.1f GB/s algorithmic = %.3f of 8 TB/s   working set %5.0f MB   %s'
          % (E, mode, ring, us, E * A / us / 1e3, byts / us / 1e3, byts / us / 1e3 / 8000.0, touched / 1e6, env.last_kernel('step')), flush=True)
    if mode == 'graph':
        graph.close()
    env.close()
    del actions, out
    torch.cuda.empty_cache()


import subprocess
if len(sys.argv) > 1:        # child: one (E, ring, mode) case under the environment the parent chose
    run(int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4]) if len(sys.argv) > 4 else 16)
    raise SystemExit(0)
for E in (4096, 16384, 65536, 131072, 262144, 524288, 1048576, 2097152, 4194304):
    run(E, 16, 'graph')
    if E >= 262144:          # the resident-grid / LDS-table form against one block per 256 lanes (MAPF_TUNE step_big=0 / 2)
        for big in ('0', '2'):
            out = subprocess.run([sys.executable, os.path.abspath(__file__), str(E), '4' if E >= 1048576 else '16', 'graph'],
                                 env=dict(os.environ, MAPF_TUNE='step_big=' + big), stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.decode()
            print('   MAPF_TUNE step_big=%s: %s' % (big, out.strip().splitlines()[-1] if out.strip() else 'failed'), flush=True)
    if E in (65536, 1048576):
        run(E, 16, 'plain')
    if E >= 1048576:
        run(E, 4, 'graph')
