#!/usr/bin/env python3
"""Long parity run of the fused rollout on the bench workload (65536 envs x 8 agents): N launches of T steps with the
streamed action ring, totals and final state against the C oracle after every launch, the full recorded trajectory
of the last launch step by step.  Not part of the test suite (about a minute of single-core oracle time per 1000 steps).

    python tools/soak_parity.py [steps=2048] [T=256] [config=c3] [envs=the config's per-GPU batch]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'gym-mapf_amd'), os.path.join(ROOT, 'oracle'), ROOT]
import bench  # noqa: E402
import c_oracle  # noqa: E402
import mapf_oracle as mo  # noqa: E402
from gym_mapf_amd.envs.vec_env import OptimizationCriteria, VecMapfEnv  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
T = int(sys.argv[2]) if len(sys.argv) > 2 else 256
CFG = bench.CONFIGS[sys.argv[3] if len(sys.argv) > 3 else 'c3']
E, A = int(sys.argv[4]) if len(sys.argv) > 4 else CFG['envs'], CFG['agents']
grid, _, nbr, start, goal = bench.workload_tables(CFG, E, 0)
bits = lambda x: np.ascontiguousarray(x).view(np.uint64)  # noqa: E731
for crit, ocrit in ((OptimizationCriteria.Makespan, mo.MAKESPAN), (OptimizationCriteria.SoC, mo.SOC)):
    env = VecMapfEnv(grid, A, None, None, CFG['fail_prob'], bench.R_CLASH, bench.R_GOAL, bench.R_LIVING, crit, seed=bench.SEED,
                     start_local=start, goal_local=goal)
    co = c_oracle.COracle(nbr, A, start, goal, CFG['fail_prob'], bench.R_CLASH, bench.R_GOAL, bench.R_LIVING, ocrit, seed=bench.SEED)
    acc, t0, done = None, time.time(), 0
    while done < steps:
        n = min(T, steps - done)
        acts = env.fill_random_actions(done, n)
        last = done + n >= steps
        acc = env.rollout(n, actions=acts, auto_reset=True, record=last, accumulate_into=acc)
        if last:
            keep = co.state.copy(), co.t
            ret = np.zeros(E)
            for t in range(n):
                ref = co.step(acts[t], auto_reset=True)
                assert np.array_equal(acc['local'][t], ref['local']) and np.array_equal(bits(acc['reward'][t]), bits(ref['reward']))
                assert np.array_equal(bits(acc['prob'][t]), bits(ref['prob'])), t
                assert np.array_equal(acc['done'][t], ref['done']) and np.array_equal(acc['collision'][t], ref['collision'])
        else:
            co.rollout(n, actions=acts, auto_reset=True)
        assert np.array_equal(env.get_state()[0], co.state), done
        done += n
        print('%s: %d steps ok (%.0f s)' % (crit.name, done, time.time() - t0), flush=True)
    print('    kernel: %s' % env.last_kernel('rollout'), flush=True)
    env.close()
print('soak parity ok: %d steps x %d envs x %d agents, both criteria' % (steps, E, A))
