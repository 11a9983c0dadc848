#!/usr/bin/env python3
"""Thread-per-env rollout kernels with SGPR spills (A = 7..16; never dispatched by the shipped library) against the C
oracle: full and ragged batches, recorded trajectory, streamed and in-kernel actions.
    MAPF_HIP_LIB=.../variants/libmapf_tpe16.so python tools/exp/tpe_spill_repro.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, 'gym-mapf_amd'), os.path.join(ROOT, 'oracle'), ROOT]
import numpy as np
import c_oracle, mapf_oracle as mo, philox
from gym_mapf_amd.envs.grid import MapfGrid
from gym_mapf_amd.envs.vec_env import OptimizationCriteria, VecMapfEnv

bits = lambda x: np.asarray(x, np.float64).view(np.uint64)
bad = 0
for A in (7, 8, 9, 12, 15, 16):
    for E in (64, 257, 1000, 4096, 16448):
        rs = np.random.RandomState(500 + A)
        lines = [''.join('@' if rs.rand() < 0.15 else '.' for _ in range(20)) for _ in range(20)]
        grid = MapfGrid(lines)
        valid, _, nbr = grid.tables()
        V, T = len(valid), 14
        start = np.argsort(rs.rand(E, V), axis=1)[:, :A].astype(np.uint16)
        goal = np.argsort(rs.rand(E, V), axis=1)[:, :A].astype(np.uint16)
        goal[::7] = start[::7]
        ids = 77 + np.arange(E)
        for crit, ocrit, auto, streamed in ((OptimizationCriteria.Makespan, mo.MAKESPAN, True, True), (OptimizationCriteria.SoC, mo.SOC, False, False)):
            env = VecMapfEnv(grid, A, None, None, 0.2, -1000.0, 100.0, -1.0, crit, seed=11, env_id_offset=77,
                             start_local=start, goal_local=goal, kernel='thread_per_env')
            co = c_oracle.COracle(nbr, A, start, goal, 0.2, -1000.0, 100.0, -1.0, ocrit, seed=11, env_id_offset=77)
            acts = np.stack([philox.random_actions_np(11, ids, t, A) for t in range(T)])
            res = env.rollout(T, actions=acts if streamed else None, auto_reset=auto, record=True)
            label = env.last_kernel('rollout')
            n_bad = 0
            for t in range(T):
                ref = co.step(acts[t], auto_reset=auto)
                wrong = (res['local'][t] != ref['local']).any(axis=1) | (bits(res['reward'][t]) != bits(ref['reward'])) | \
                        (bits(res['prob'][t]) != bits(ref['prob'])) | (res['done'][t] != ref['done']) | (res['collision'][t] != ref['collision'])
                if wrong.any() and n_bad == 0:
                    print('   first mismatch: A=%d E=%d t=%d envs %s' % (A, E, t, np.nonzero(wrong)[0][:16].tolist()))
                n_bad += int(wrong.sum())
            n_bad += int((env.get_state()[0] != co.state).any(axis=1).sum())
            bad += n_bad
            print('A=%2d E=%6d %-8s auto=%d streamed=%d: %s  [%s]' % (A, E, crit.value, auto, streamed, 'OK' if n_bad == 0 else '%d WRONG env-steps' % n_bad, label))
            env.close()
print('total wrong env-steps:', bad)
