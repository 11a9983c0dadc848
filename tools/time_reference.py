#!/usr/bin/env python3
"""Time the REAL reference (read-only at /root/reference) in the build container on BASELINE configs C1-C3
(CPU baseline plan step 1): one process, one core, warm caches, best of 3 repetitions.

    python tools/time_reference.py            # writes profiles/reference_cpu.json + profiles/r02_reference_cpu.txt

The reference cannot travel to the GPU box, so bench.py's cpu_baseline there times the pure-Python port
(oracle/mapf_oracle.py, kind "port") and quotes these committed figures beside it.  The reference is imported the way
tests/golden/make_golden.py does (in-memory stand-ins for its absent gym / colorama imports); its step() is untouched
and draws from its own np_random (seeded MT19937) here -- this is a timing, not a parity run.
"""
import json
import os
import platform
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import make_golden as mg  # noqa: E402  (installs the stand-ins and imports the reference)

R_CLASH, R_GOAL, R_LIVING = -1000.0, 100.0, -1.0


def time_config(map_name, scen_id, n_agents, fail_prob, n_steps, reps=3):
    best = None
    for _ in range(reps):
        env = mg.create_mapf_env(map_name, scen_id, n_agents, fail_prob, R_CLASH, R_GOAL, R_LIVING, mg.CRITERIA['Makespan'])
        rng = random.Random(0)
        acts = [rng.randrange(env.nA) for _ in range(n_steps)]
        for a in acts[:2000]:                      # warm the lru_caches the way a long run would
            if env.step(a)[2]:
                env.reset()
        t0 = time.perf_counter()
        for a in acts:
            if env.step(a)[2]:
                env.reset()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return n_steps / best


def main():
    rows = {}
    for key, (name, scen, A, fp, n) in {'c1': ('empty-8-8', 1, 2, 0.0, 50000), 'c2': ('empty-16-16', 1, 4, 0.1, 50000),
                                        'c3': ('room-32-32-4', 6, 8, 0.2, 50000)}.items():
        rate = time_config(name, scen, A, fp, n)
        rows[key] = {'map': name, 'scen_id': scen, 'n_agents': A, 'fail_prob': fp, 'env_steps': n,
                     'env_steps_per_s': rate, 'agent_steps_per_s': rate * A}
        print('%s %-13s A=%d slip=%.1f: %9.0f env-steps/s = %9.0f agent-steps/s' % (key, name, A, fp, rate, rate * A))
    out = {'what': 'gym_mapf.envs.mapf_env.MapfEnv.step() of the unmodified reference, one env, one process, 1 core, '
                   'random joint actions, reset on done, best of 3 x 50000 steps after 2000 warm-up steps',
           'host': '%s, %d vCPUs (build container)' % (platform.processor() or platform.machine(), os.cpu_count()),
           'python': platform.python_version(), **rows}
    with open(os.path.join(ROOT, 'profiles', 'reference_cpu.json'), 'w') as f:
        json.dump(out, f, indent=1)
    with open(os.path.join(ROOT, 'profiles', 'r02_reference_cpu.txt'), 'w') as f:
        f.write(out['what'] + '\n' + out['host'] + ', python ' + out['python'] + '\n')
        for k in ('c1', 'c2', 'c3'):
            r = rows[k]
            f.write('%s %-13s A=%d slip=%.1f: %9.0f env-steps/s = %9.0f agent-steps/s\n'
                    % (k, r['map'], r['n_agents'], r['fail_prob'], r['env_steps_per_s'], r['agent_steps_per_s']))


if __name__ == '__main__':
    main()
