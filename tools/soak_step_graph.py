#!/usr/bin/env python3
"""Long parity run of the recorded single step: N mapf_step calls recorded once into a hipGraph (device arrays, state
view, auto-reset) and replayed R times on the bench workload; after every replay the state view, the step index and
every node's outputs of that replay against the C oracle.  Not part of the test suite.

    python tools/soak_step_graph.py [replays=32] [nodes=32] [config=c3] [envs=the config's per-GPU batch]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'gym-mapf_amd'), os.path.join(ROOT, 'oracle'), ROOT]
import torch  # noqa: E402,F401
import bench  # noqa: E402
import c_oracle  # noqa: E402
import mapf_oracle as mo  # noqa: E402
from gym_mapf_amd.envs.vec_env import OptimizationCriteria, VecMapfEnv  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N = int(sys.argv[2]) if len(sys.argv) > 2 else 32
CFG = bench.CONFIGS[sys.argv[3] if len(sys.argv) > 3 else 'c3']
E, A = int(sys.argv[4]) if len(sys.argv) > 4 else CFG['envs'], CFG['agents']
grid, _, nbr, start, goal = bench.workload_tables(CFG, E, 0)
bits = lambda x: np.ascontiguousarray(x).view(np.uint64)  # noqa: E731
for crit, ocrit in ((OptimizationCriteria.Makespan, mo.MAKESPAN), (OptimizationCriteria.SoC, mo.SOC)):
    env = VecMapfEnv(grid, A, None, None, CFG['fail_prob'], bench.R_CLASH, bench.R_GOAL, bench.R_LIVING, crit, seed=bench.SEED,
                     device_arrays=True, start_local=start, goal_local=goal)
    co = c_oracle.COracle(nbr, A, start, goal, CFG['fail_prob'], bench.R_CLASH, bench.R_GOAL, bench.R_LIVING, ocrit, seed=bench.SEED)
    actions = env.fill_random_actions(0, N)
    env.sync()
    acts = actions.cpu().numpy()
    env.graph_begin()
    outs = []
    for k in range(N):
        call, out = env.prepare_step(actions[k], auto_reset=True, write_local=False)
        call()
        outs.append(out)
    graph = env.graph_end()
    view = env.state_view()
    t0, episodes = time.time(), 0
    for rep in range(R):
        graph.launch(1)
        env.sync()
        for k in range(N):
            ref = co.step(acts[k], auto_reset=True)
            o = outs[k]
            assert np.array_equal(bits(o['reward'].cpu().numpy()), bits(ref['reward'])) and np.array_equal(bits(o['prob'].cpu().numpy()), bits(ref['prob'])), (rep, k)
            assert np.array_equal(o['done'].cpu().numpy(), ref['done']) and np.array_equal(o['collision'].cpu().numpy(), ref['collision']), (rep, k)
            assert np.array_equal(o['was_terminal'].cpu().numpy(), ref['was_terminal']), (rep, k)
            episodes += int(ref['done'].sum())
        assert np.array_equal(view.cpu().numpy(), co.state) and env.t == co.t == (rep + 1) * N, rep
        if rep % 8 == 7:
            print('%s: %d replays = %d steps ok, %d episodes ended (%.0f s)' % (crit.name, rep + 1, (rep + 1) * N, episodes, time.time() - t0), flush=True)
    print('    kernel: %s' % env.last_kernel('step'), flush=True)
    graph.close()
    env.close()
print('recorded-step soak ok: %d replays x %d nodes x %d envs x %d agents, both criteria' % (R, N, E, A))
