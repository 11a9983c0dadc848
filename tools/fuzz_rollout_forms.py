#!/usr/bin/env python3
"""Differential fuzz of the fused rollout: the packed kernels (whatever form the dispatch picks) against the pair-layout
lane-group kernel (MAPF_TUNE=quad_lanes=0) on random shapes -- team size, batch, launch lengths and phases, slip, criteria,
auto-reset, streamed actions or the in-kernel policy, recorded or totals only.  Both run on the GPU, so hundreds of cases take a
minute; every recorded array, the totals and the final state must agree bit for bit.  Not part of the suite.
    python3 tools/fuzz_rollout_forms.py [cases=200] [seed=0]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'gym-mapf_amd'), os.path.join(ROOT, 'oracle'), ROOT]
import philox  # noqa: E402
from gym_mapf_amd.envs.grid import MapfGrid  # noqa: E402
from gym_mapf_amd.envs.vec_env import OptimizationCriteria, VecMapfEnv  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bits = lambda x: np.ascontiguousarray(x).view(np.uint8)  # noqa: E731
kernels = {}
for case in range(cases):
    A = int(rs.choice([4, 8, 8, 16, 32]))
    side = int(rs.choice([10, 16, 24, 32] if A <= 16 else [24, 40, 64]))
    grid = MapfGrid([''.join('@' if rs.rand() < 0.12 else '.' for _ in range(side)) for _ in range(side)])
    V = len(grid.tables()[0])
    E = int(rs.choice([1024, 2048, 4096, 8192])) * (2 if A <= 8 and rs.rand() < 0.3 else 1)
    crit = OptimizationCriteria.SoC if rs.rand() < 0.4 else OptimizationCriteria.Makespan
    fail = float(rs.choice([0.0, 0.1, 0.2, 0.5, 1.0]))
    auto, streamed, record = bool(rs.rand() < 0.7), bool(rs.rand() < 0.6), bool(rs.rand() < 0.7)
    start = np.argsort(rs.rand(E, V), axis=1)[:, :A].astype(np.uint16)
    goal = np.argsort(rs.rand(E, V), axis=1)[:, :A].astype(np.uint16)
    goal[::5] = start[::5]
    seed, t0 = int(rs.randint(1, 1000)), int(rs.randint(0, 9))
    lengths = [int(rs.randint(1, 40)) for _ in range(int(rs.randint(1, 4)))]

    def run(tune):
        if tune:
            os.environ['MAPF_TUNE'] = tune
        else:
            os.environ.pop('MAPF_TUNE', None)
        env = VecMapfEnv(grid, A, None, None, fail, -1000.0, 100.0, -1.0, crit, seed=seed, start_local=start, goal_local=goal)
        env.set_state(None, t=t0)
        outs, t = [], t0
        ids = np.arange(E)
        for n in lengths:
            acts = np.stack([philox.random_actions_np(seed + 5, ids, t + j, A) for j in range(n)]) if streamed else None
            res = env.rollout(n, actions=acts, auto_reset=auto, record=record)
            outs.append({k: np.array(v) for k, v in res.items()})
            t += n
        name = env.last_kernel('rollout')
        state = env.get_state()
        env.close()
        return outs, state, name

    a, sa, ka = run(None)
    b, sb, kb = run('quad_lanes=0')
    tag = (case, A, E, side, crit, fail, auto, streamed, record, t0, lengths, ka, kb)
    assert 'lq_rollout_kernel' not in kb, tag
    if 'lq_rollout_kernel' not in ka:                             # (no packed form for this shape: nothing to compare)
        kernels['(lane-group kernel both times)'] = kernels.get('(lane-group kernel both times)', 0) + 1
        continue
    kernels[ka.split('>')[0] + '>'] = kernels.get(ka.split('>')[0] + '>', 0) + 1
    assert sa[1] == sb[1] and np.array_equal(sa[0], sb[0]), tag
    for x, y in zip(a, b):
        for k in y:
            assert np.array_equal(bits(x[k]), bits(y[k])), (k,) + tag
    if case % 20 == 19:
        print('%d cases ok' % (case + 1), flush=True)
print('fuzz ok: %d cases, packed kernels seen:' % cases)
for k, v in sorted(kernels.items()):
    print('  %4d  %s' % (v, k))
