#!/usr/bin/env python3
"""Throughput of the fused rollout (recorded trajectory, streamed actions) and of single-step launches on the
BASELINE.json configurations other than the headline one -- informational, not the bench line.

    python tools/bench_configs.py            # on a MI355X
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'gym-mapf_amd'), ROOT]
import torch  # noqa: E402,F401  (initialise torch's HIP runtime first)
from gym_mapf_amd.envs import map_name_to_files  # noqa: E402
from gym_mapf_amd.envs.grid import MapfGrid  # noqa: E402
from gym_mapf_amd.envs.utils import parse_map_file, parse_scen_file  # noqa: E402
from gym_mapf_amd.envs.vec_env import OptimizationCriteria, VecMapfEnv  # noqa: E402


def scen_tables(map_name, scen_ids, A, E):
    grid = MapfGrid(parse_map_file(map_name_to_files(map_name, scen_ids[0])[0]))
    _, l2i, _ = grid.tables()
    per = [parse_scen_file(map_name_to_files(map_name, sid)[1], A) for sid in scen_ids]
    which = np.arange(E) % len(scen_ids)
    start = np.asarray([[l2i[l] for l in p[0]] for p in per], np.uint16)[which]
    goal = np.asarray([[l2i[l] for l in p[1]] for p in per], np.uint16)[which]
    return grid, np.ascontiguousarray(start), np.ascontiguousarray(goal)


def random_tables(E, A, size=64, p_obst=0.2):
    rs = np.random.RandomState(20)
    obst = rs.rand(size, size) < p_obst
    grid = MapfGrid([''.join('@' if obst[r, c] else '.' for c in range(size)) for r in range(size)])
    V = len(grid.tables()[0])
    r2 = np.random.RandomState(42)
    start = np.argsort(r2.rand(E, V), axis=1)[:, :A].astype(np.uint16)
    goal = np.argsort(r2.rand(E, V), axis=1)[:, :A].astype(np.uint16)
    return grid, start, goal


def measure(name, grid, start, goal, A, fail_prob, T=64, reps=12):
    E = start.shape[0]
    env = VecMapfEnv(grid, A, None, None, fail_prob, -1000.0, 100.0, -1.0, OptimizationCriteria.Makespan, seed=42,
                     device_arrays=True, start_local=start, goal_local=goal)
    actions = env.fill_random_actions(0, T)
    env.rollout(T, actions=actions, auto_reset=True, record=True)
    env.sync()
    env.timer_begin()
    for _ in range(reps):
        env.rollout(T, actions=actions, auto_reset=True, record=True)
    ms = env.timer_end()
    ro = reps * T * E * A / (ms * 1e-3)
    calls = [env.prepare_step(actions[k], auto_reset=True)[0] for k in range(T)]
    for c in calls[:8]:
        c()
    env.sync()
    env.timer_begin()
    for r in range(4):
        for c in calls:
            c()
    ms1 = env.timer_end()
    st = 4 * T * E * A / (ms1 * 1e-3)
    bpas = 5.0 + 18.0 / A
    print('%-46s E=%7d A=%3d V=%5d | rollout %7.1f G agent-steps/s (%.3f of 8 TB/s) | single-step %6.1f G (%.3f) | %s | %s' % (
        name, E, A, len(grid.tables()[0]), ro / 1e9, ro * bpas / 8e12, st / 1e9, st * bpas / 8e12,
        env.last_kernel('rollout').split(' (')[0], env.last_kernel('step').split(' (')[0]), flush=True)
    env.close()


if __name__ == '__main__':
    g, s, t = scen_tables('empty-16-16', list(range(1, 26)), 4, 4096)
    measure('C2 empty-16-16, 4 agents, slip 0.1', g, s, t, 4, 0.1)
    g, s, t = scen_tables('room-32-32-4', [6, 12, 13, 23, 24, 25], 8, 65536)
    measure('C3 room-32-32-4, 8 agents, slip 0.2 (bench)', g, s, t, 8, 0.2)
    g, s, t = scen_tables('room-32-32-4', [6, 12, 13, 23, 24, 25], 8, 32768)
    measure('C4 share: room-32-32-4, 8 agents, 32768/GPU', g, s, t, 8, 0.2)
    g, s, t = random_tables(16384, 32)
    measure('C5 share: random-64-64-20*, 32 agents, 16384/GPU', g, s, t, 32, 0.2)
    g, s, t = scen_tables('room-64-64-16', [1, 2, 5, 7], 32, 16384)
    measure('room-64-64-16, 32 agents, 16384 envs', g, s, t, 32, 0.2)
    # the reference's large maps: move tables of 1.2 MB / 3.8 MB, i.e. the global-table (L2-resident) kernels
    g, s, t = scen_tables('maze-128-128-10', [18], 32, 16384)
    measure('maze-128-128-10, 32 agents (scen 18), 16384 envs', g, s, t, 32, 0.2)
    g, s, t = scen_tables('Berlin_1_256', [11], 4, 65536)
    measure('Berlin_1_256, 4 agents (scen 11), 65536 envs', g, s, t, 4, 0.2)
    g, s, t = scen_tables('Berlin_1_256', [2, 4, 8, 11, 14, 18, 22, 24], 2, 65536)
    measure('Berlin_1_256, 2 agents (8 scens), 65536 envs', g, s, t, 2, 0.1)
    g, s, t = scen_tables('empty-8-8', [1], 2, 65536)
    measure('empty-8-8, 2 agents, slip 0 (C1 map), 65536 envs', g, s, t, 2, 0.0)
