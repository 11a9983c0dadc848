#!/usr/bin/env python3
"""bench.py -- agent-steps/s of the batched MapfEnv.step() hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c3|c4|c5|c2]

N > 1 without a launcher: this process starts N children itself (one rank per GPU, RCCL), before it touches
any GPU; under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` it is one of the ranks.

Workloads (BASELINE.json configs):
  c3 (default)  configs[2]: room-32-32-4, 8 agents, slip 0.2, 65536 envs PER GPU (weak scaling; env e uses scen id
                {6,12,13,23,24,25}[e mod 6])
  c4            configs[3]: the same map, 262144 envs IN TOTAL split evenly over the N GPUs (strong scaling)
  c5            configs[4]: synthetic random-64-64-20 map (the MovingAI file is not shipped: 64x64, cells blocked
                with p = 0.2 from RandomState(20)), 32 agents, slip 0.2, 131072 envs in total over the N GPUs,
                per-env seeded random distinct start / goal cells
  c2            configs[1]: empty-16-16, 4 agents, slip 0.1, 4096 envs per GPU (env e uses scen id 1 + e mod 25)
Global env ids feed the RNG counters, so results do not depend on the rank count.  Actions are synthetic
(uniform-random policy stream), resident in HBM before the timed region; every done env is auto-reset as the
reference's caller loop does.

A CLI "step" is ONE PASS of the hot path over the rank's batch as the library runs it at speed: one mapf_rollout
launch = T = 256 fused MapfEnv.step() calls of every env (`config.env_steps_per_step`), each of which writes all
its outputs (next cells, reward, done, collision, prob) to HBM.  K steps = K launches back to back on the env's
HIP stream between barrier + synchronize on both sides; `value` = all ranks' agent-steps / max-over-ranks wall
time of that region.  Rank 0 prints ONE JSON line.  Keys beyond the contract:
  value / ms_per_step   MEDIAN over --repeats timed blocks of exactly K steps each (every block bracketed by barrier +
                synchronize on both sides, max over ranks); `repeats` lists every block.  `value_cold` = the first 25
                launches after >= 1 s of idle device, timed the same way before any warm-up (clocks not yet ramped)
  roofline      dominant kernel (the fused rollout; `kernel` = what the library reports it dispatched): algorithmic
                bytes / HIP-event time per launch; `traffic` = HBM bytes per launch from the committed PMC passes
                (profiles/traffic.json: per-step + fixed part), `traffic_frac` = that / the same time / peak;
                `valu_frac` = the time a SIMD's vector ALU needs for its share of the launch's vector instructions -- the
                committed SQ counter passes' instruction counts by kind (profiles/valu.json: 64-bit integer, float64
                multiply / add as counted, the 32-bit rest split by the kernel's static mix into fast and ordinary
                encodings) x the measured issue time of each kind (profiles/r04_valu_issue_cost*.txt) -- divided by the
                live launch time; traffic / valu entries made from other kernel sources than this tree's (csrc_hash) are
                not used: the keys then print null
  single_step_launches   the same env-steps as one mapf_step launch each, recorded ONCE into a hipGraph (256 nodes; the
                step index lives in device memory, so every replay draws fresh numbers) and replayed; the next
                observation is read from the handle's state view (cells written once); `plain_launches` = the same
                calls issued one by one from the host (host-enqueue bound)
  per_gpu_shapes   the one-GPU shares of configs[3] (32768 envs x 8 agents) and configs[4] (16384 envs x 32 agents): the same
                fused launch, HIP-event time of 3 x 10 launches each (default configuration only)
  policy_rollout   the headline launch with actions = NULL: the in-kernel policy stream (one Philox call per agent quad per four
                steps, a byte per action) stands in for the caller's `a = policy(s)`; same recording, same roofline contract
  baseline_configs   multi-rank runs only (and --force-dist): configs[3] (262144 envs x 8 agents) and configs[4] (131072 x 32)
                sharded over THIS run's ranks in block-aligned shards -- per config: value (all ranks' agent-steps / max-over-
                ranks time), rank 0's roofline.frac, kernel, shards, the gather of the returns, rank 0's shard against the C oracle
  transitions   env.P[s][a] (mapf_transitions_compact): branches/s and written bytes / HIP-event time of 8 agents x 20000 and
                4 agents x 2 M random queries on room-32-32-4 (2A + 18 bytes per branch; an HBM-write-bound kernel by contract)
  scalar_env    the reference's own regime (configs[0]: empty-8-8, 2 agents, slip 0, ONE env): MapfEnv.step()
                calls per second through the drop-in class, beside the reference's build-container figure
  cpu_baseline  the pure-Python restatement of the reference (oracle/, kind "port") timed on this box's host
                cores on a bounded sample (rank 0, N=1 only), the real reference's build-container rate beside it;
                `all_cores` = N = usable host cores independent processes of the C port and of the Python port (summed rates:
                the whole-host figure; measured before this process touches the GPU)
  parity        bit-exact check of the first steps of this very run against the C oracle
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, 'gym-mapf_amd'))

import numpy as np  # noqa: E402

R_CLASH, R_GOAL, R_LIVING = -1000.0, 100.0, -1.0
SEED = 42
SHARD_GRANULE = 1024           # strong-scaling shards are multiples of this many envs (whole blocks of every packed kernel form)
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.3 TB/s achievable)

CONFIGS = {
    'c2': dict(baseline='configs[1]', map='empty-16-16', scen_ids=tuple(range(1, 26)), agents=4, fail_prob=0.1,
               envs=4096, scaling='weak'),
    'c3': dict(baseline='configs[2]', map='room-32-32-4', scen_ids=(6, 12, 13, 23, 24, 25), agents=8, fail_prob=0.2,
               envs=65536, scaling='weak'),
    'c4': dict(baseline='configs[3]', map='room-32-32-4', scen_ids=(6, 12, 13, 23, 24, 25), agents=8, fail_prob=0.2,
               envs=262144, scaling='strong'),
    'c5': dict(baseline='configs[4]', map='random-64-64-20 (synthetic stand-in)', scen_ids=None, agents=32,
               fail_prob=0.2, envs=131072, scaling='strong'),
}


def bytes_per_agent_step(A):
    """u16 state in + u8 action + u16 state out per agent; f64 reward + f64 prob +
    u8 done + u8 collision per env."""
    return 5.0 + 18.0 / A


_CSRC_HASH = None


def csrc_hash():
    """Identity of the kernel sources a committed counter profile belongs to: sha256 (first 16 hex digits) over
    gym-mapf_amd/csrc/*.hip (but mapf_transitions.hip), *.hpp and include/mapf_hip.h with comments and white space removed.  profiles/traffic.json
    and profiles/valu.json entries carry it (tools/derive_traffic.py, tools/derive_valu.py); an entry made from other
    sources than the ones in this tree is NOT used -- `roofline.traffic` / `valu_frac` then print null instead of a stale
    number that merely shares the kernel's label."""
    global _CSRC_HASH
    if _CSRC_HASH is None:
        import glob
        import hashlib
        import re
        csrc = os.path.join(ROOT, 'gym-mapf_amd', 'csrc')
        # (mapf_transitions.hip holds the env.P kernels only -- no traffic / valu entry describes them -- so editing it does not
        # orphan the rollout / step kernels' counter profiles)
        files = sorted(f for f in glob.glob(os.path.join(csrc, '*.hip')) + glob.glob(os.path.join(csrc, '*.hpp'))
                       if os.path.basename(f) != 'mapf_transitions.hip') + [os.path.join(ROOT, 'include', 'mapf_hip.h')]
        h = hashlib.sha256()
        for path in files:
            with open(path) as f:
                text = f.read()
            text = re.sub(r'/\*.*?\*/', ' ', text, flags=re.S)
            text = re.sub(r'//[^\n]*', ' ', text)
            h.update(os.path.basename(path).encode())
            h.update(''.join(text.split()).encode())
        _CSRC_HASH = h.hexdigest()[:16]
    return _CSRC_HASH


def measured_traffic(kernel, n_envs, n_agents, steps_per_launch):
    """HBM bytes per launch of `kernel` at this batch, from the committed rocprofv3 PMC passes of this same command
    (profiles/traffic.json: FETCH_SIZE and WRITE_SIZE collected in separate passes, FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for gfx950; two launch lengths give a per-step and a fixed part).  None when no
    profile of this kernel instance at this batch is committed."""
    try:
        with open(os.path.join(ROOT, 'profiles', 'traffic.json')) as f:
            entries = json.load(f)['kernels']
    except (OSError, KeyError, ValueError):
        return None
    for entry in entries if isinstance(entries, list) else ():
        if entry.get('kernel') == kernel and entry.get('n_envs') == n_envs and entry.get('n_agents') == n_agents and \
                entry.get('csrc_hash') == csrc_hash():
            return entry['fixed_bytes'] + entry['bytes_per_env_step_launch'] * steps_per_launch
    return None


def synthetic_random_map(size=64, p_obst=0.2, seed=20):
    """the stand-in for the unshipped random-64-64-20 file."""
    obst = np.random.RandomState(seed).rand(size, size) < p_obst
    return [''.join('@' if obst[r, c] else '.' for c in range(size)) for r in range(size)]


def random_distinct_cells(n_cells, n_agents, env_ids, salt):
    """[len(env_ids), A] distinct free cells per env, a function of the GLOBAL env id only (chunks of 4096 ids share
    one RandomState seeded by (SEED, salt, chunk); rows with a repeated cell are redrawn until none is left)."""
    env_ids = np.asarray(env_ids, dtype=np.int64)
    out = np.empty((len(env_ids), n_agents), np.uint16)
    chunk = 4096
    for c in np.unique(env_ids // chunk):
        rs = np.random.RandomState([SEED, salt, int(c)])
        cells = rs.randint(0, n_cells, size=(chunk, n_agents))
        while True:
            srt = np.sort(cells, axis=1)
            bad = np.nonzero((srt[:, 1:] == srt[:, :-1]).any(axis=1))[0]
            if bad.size == 0:
                break
            cells[bad] = rs.randint(0, n_cells, size=(bad.size, n_agents))
        sel = np.nonzero(env_ids // chunk == c)[0]
        out[sel] = cells[env_ids[sel] % chunk]
    return out


def workload_tables(cfg, n_envs, env_id_offset):
    """(grid, map lines, nbr, start u16[E, A], goal u16[E, A]) of this rank's slice of the configuration."""
    from gym_mapf_amd.envs import map_name_to_files
    from gym_mapf_amd.envs.grid import MapfGrid
    from gym_mapf_amd.envs.utils import parse_map_file, parse_scen_file
    A = cfg['agents']
    ids = env_id_offset + np.arange(n_envs)
    if cfg['scen_ids'] is None:
        lines = synthetic_random_map()
        grid = MapfGrid(lines)
        V = len(grid.tables()[0])
        start = random_distinct_cells(V, A, ids, 1)
        goal = random_distinct_cells(V, A, ids, 2)
        return grid, lines, grid.tables()[2], start, goal
    lines = parse_map_file(map_name_to_files(cfg['map'], cfg['scen_ids'][0])[0])
    grid = MapfGrid(lines)
    _, loc_to_int, nbr = grid.tables()
    per_scen = []
    for sid in cfg['scen_ids']:
        s, g = parse_scen_file(map_name_to_files(cfg['map'], sid)[1], A)
        per_scen.append(([loc_to_int[l] for l in s], [loc_to_int[l] for l in g]))
    which = ids % len(cfg['scen_ids'])
    start = np.asarray([p[0] for p in per_scen], np.uint16)[which]
    goal = np.asarray([p[1] for p in per_scen], np.uint16)[which]
    return grid, lines, nbr, np.ascontiguousarray(start), np.ascontiguousarray(goal)


def reference_cpu_figures():
    """The REAL reference timed in the build container (tools/time_reference.py -> profiles/reference_cpu.json; the
    reference may not travel to the GPU box, so this is a committed measurement, not re-run here)."""
    try:
        with open(os.path.join(ROOT, 'profiles', 'reference_cpu.json')) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


def cpu_baseline(cfg, budget_s=12.0):
    """Time the CPU restatements of the reference on this box (bounded sample of the same workload).

    value = the pure-Python scalar port (oracle/mapf_oracle.py OracleEnv: one env object, Python
    loops and tuples like the reference), 1 core.  The plain-C port's rate is reported beside it."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import c_oracle
    import mapf_oracle as mo
    import philox
    A = cfg['agents']
    grid, lines, nbr, start, goal = workload_tables(cfg, 16384, 0)
    valid = grid.tables()[0]
    env = mo.OracleEnv(lines, A, [valid[c] for c in start[0]], [valid[c] for c in goal[0]], cfg['fail_prob'],
                       R_CLASH, R_GOAL, R_LIVING, mo.MAKESPAN)
    chunk = 2000
    acts = [philox.random_actions_np(SEED, [0], t, A)[0].tolist() for t in range(chunk)]
    us = [philox.slip_uniforms_np(SEED, [0], t, A)[0].tolist() for t in range(chunk)]
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        for a, u in zip(acts, us):
            _, _, done, _, _, _ = env.step(a, u)
            if done:
                env.reset()
        n += chunk
    py_rate = n * A / (time.perf_counter() - t0)

    E = 16384
    co = c_oracle.COracle(nbr, A, start, goal, cfg['fail_prob'], R_CLASH, R_GOAL, R_LIVING, mo.MAKESPAN, seed=SEED)
    co.rollout(4)
    t0 = time.perf_counter()
    steps = 0
    while time.perf_counter() - t0 < 3.0:
        co.rollout(16)
        steps += 16
    c_rate = steps * E * A / (time.perf_counter() - t0)
    out = {"value": py_rate, "unit": "agent-steps/s", "cores": 1, "kind": "port",
           "sample": "oracle/mapf_oracle.py OracleEnv (pure-Python restatement of MapfEnv.step), %s, env 0's scenario, "
                     "%d agents, slip %g, one env, %d env-steps with reset on done, 1 core of %d"
                     % (cfg['map'], A, cfg['fail_prob'], n, os.cpu_count()),
           "c_port_value": c_rate,
           "c_port_sample": "oracle/mapf_oracle.c scalar C, %d envs x %d steps, 1 core" % (E, steps)}
    ref = reference_cpu_figures()
    if ref:
        out["reference_build_container"] = ref
    return out


def usable_cores():
    """Host cores this process may actually use: os.cpu_count() cut down by the affinity mask and the cgroup's CPU quota (a GPU
    box hands one GPU's job a share of the host, not all of it)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):
        try:
            with open(path) as f:
                fields = f.read().split()
            if path.endswith('cpu.max'):
                if fields[0] != 'max':
                    n = min(n, max(1, int(float(fields[0]) / float(fields[1]) + 0.999)))
            else:
                quota = int(fields[0])
                if quota > 0:
                    with open('/sys/fs/cgroup/cpu/cpu.cfs_period_us') as f:
                        n = min(n, max(1, int(quota / float(f.read().split()[0]) + 0.999)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def cpu_worker(kind, config, seconds, index):
    """One process of the whole-host CPU figure (python bench.py --cpu-worker KIND CONFIG SECONDS INDEX): the C port (kind 'c')
    or the pure-Python port ('py') of the reference's step on its own slice of the workload, for `seconds`; prints
    "<agent-steps> <elapsed seconds>"."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import mapf_oracle as mo
    cfg = CONFIGS[config]
    A = cfg['agents']
    if kind == 'c':
        import c_oracle
        E = 4096
        grid, lines, nbr, start, goal = workload_tables(cfg, E, index * E)
        co = c_oracle.COracle(nbr, A, start, goal, cfg['fail_prob'], R_CLASH, R_GOAL, R_LIVING, mo.MAKESPAN, seed=SEED, env_id_offset=index * E)
        co.rollout(4)
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            co.rollout(16)
            n += 16 * E * A
    else:
        import philox
        grid, lines, nbr, start, goal = workload_tables(cfg, 1, index)
        valid = grid.tables()[0]
        env = mo.OracleEnv(lines, A, [valid[c] for c in start[0]], [valid[c] for c in goal[0]], cfg['fail_prob'], R_CLASH, R_GOAL, R_LIVING, mo.MAKESPAN)
        chunk = 1000
        acts = [philox.random_actions_np(SEED, [index], t, A)[0].tolist() for t in range(chunk)]
        us = [philox.slip_uniforms_np(SEED, [index], t, A)[0].tolist() for t in range(chunk)]
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            for a, u in zip(acts, us):
                if env.step(a, u)[2]:
                    env.reset()
            n += chunk * A
    print(n, time.perf_counter() - t0, flush=True)


def cpu_all_cores(config, seconds_c=4.0, seconds_py=6.0):
    """The whole-host figure north_star words as "the same box's host cores": N = usable_cores() independent processes of the
    plain-C port, then of the pure-Python port (the reference has no internal parallelism, so N processes is what its user
    would run), each on its own envs of the same workload, for a bounded time.  MUST run before this process touches the GPU
    (the workers are child programs).  Returns the dict that goes into cpu_baseline['all_cores']."""
    n = usable_cores()
    out = {"cores": n, "host_cpu_count": os.cpu_count(), "unit": "agent-steps/s"}
    for kind, seconds, key in (('c', seconds_c, 'c_port'), ('py', seconds_py, 'python_port')):
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), '--cpu-worker', kind, config, str(seconds), str(i)],
                                  stdout=subprocess.PIPE, stderr=subprocess.DEVNULL) for i in range(n)]
        total = 0.0
        for p in procs:
            try:
                txt = p.communicate(timeout=seconds * 5 + 120)[0].decode().split()
                total += float(txt[0]) / float(txt[1])
            except Exception:
                p.kill()
        out[key] = total
    out["sample"] = ("%d processes x (%g s of oracle/mapf_oracle.c on 4096 envs each, then %g s of oracle/mapf_oracle.py on one env each), "
                     "%s workload; summed rates" % (n, seconds_c, seconds_py, CONFIGS[config]['map']))
    return out


def scalar_env_rate(budget_s=2.0):
    """BASELINE configs[0]: empty-8-8, 2 agents, slip 0, ONE env, stepped through the drop-in MapfEnv class the way
    the reference's users do (joint-integer action in, (s, r, done, info) out, reset on done)."""
    import random
    from gym_mapf_amd.envs.utils import create_mapf_env
    from gym_mapf_amd.envs.vec_env import OptimizationCriteria
    env = create_mapf_env('empty-8-8', 1, 2, 0.0, R_CLASH, R_GOAL, R_LIVING, OptimizationCriteria.Makespan)
    rng = random.Random(0)
    acts = [rng.randrange(env.nA) for _ in range(4096)]
    for a in acts[:64]:
        if env.step(a)[2]:
            env.reset()
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        for a in acts[:512]:
            if env.step(a)[2]:
                env.reset()
        n += 512
    rate = n / (time.perf_counter() - t0)
    env.close()
    out = {"value": rate, "unit": "env-steps/s", "config": "BASELINE configs[0]: empty-8-8, 2 agents, slip 0, 1 env, "
           "MapfEnv.step() through the C ABI (pinned staging, one launch + one sync per call)", "env_steps": n}
    ref = reference_cpu_figures()
    if ref and 'c1' in ref:
        out["reference_build_container_env_steps_per_s"] = ref['c1'].get('env_steps_per_s')
    return out


def measured_valu(kernel, n_envs, n_agents, steps_per_launch):
    """The committed SQ counter passes of `kernel` at this batch (profiles/valu.json, tools/derive_valu.py): (VALU
    wave-instructions per launch, the time in ms a SIMD's vector ALU needs for its share of them at the measured issue
    rates of their kinds -- profiles/r04_valu_issue_cost*.txt), or None when no such pass of THESE kernel sources is committed."""
    try:
        with open(os.path.join(ROOT, 'profiles', 'valu.json')) as f:
            entries = json.load(f)['kernels']
    except (OSError, KeyError, ValueError):
        return None
    for entry in entries:
        if entry.get('kernel') == kernel and entry.get('n_envs') == n_envs and entry.get('n_agents') == n_agents and \
                entry.get('env_steps_per_launch') == steps_per_launch and entry.get('csrc_hash') == csrc_hash():
            return entry['valu_insts_per_launch'], entry['valu_ms_per_simd']
    return None


def spawn_ranks(n, deadline_s, command=None):
    """`python bench.py --gpus N` with no launcher: start one child per GPU (before this process makes any GPU
    call -- it makes none at all), wait for them, and exit with the first failure's code.  Rank 0 prints the line.
    A rank that has not finished `deadline_s` seconds after the start (stuck in a collective, a hung device) takes
    every rank down with it: exit code 124.  (`command` replaces the rank's command line: tests.)"""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        procs.append(subprocess.Popen(command or [sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    pending = list(procs)
    t_end = time.monotonic() + deadline_s
    while pending:
        for p in list(pending):
            code = p.poll()
            if code is None:
                continue
            pending.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in pending:      # a dead rank would leave the others waiting in a collective
                    q.terminate()
        if pending and time.monotonic() > t_end:
            sys.stderr.write('bench.py: %d rank(s) still running after %.0f s -- terminating all ranks\n' % (len(pending), deadline_s))
            for q in pending:
                q.terminate()
            t_kill = time.monotonic() + 10.0
            while any(q.poll() is None for q in pending) and time.monotonic() < t_kill:
                time.sleep(0.1)
            for q in pending:
                if q.poll() is None:
                    q.kill()
            raise SystemExit(124)
        time.sleep(0.05)
    raise SystemExit(rc)


def per_gpu_shape_rate(name, n_envs, T=256, n_launch=10, blocks=3, preroll_ms=60.0, policy=False):
    """The fused rollout on what ONE GPU runs of a BASELINE configuration that is sharded over eight (configs[3]: 32768 envs
    of 8 agents, configs[4]: 16384 envs of 32 agents) -- the same launch as the headline leg (T env-steps per launch, every
    env-step's outputs recorded to HBM, actions streamed from a two-slot ring), HIP events around `n_launch` launches on the
    handle's stream, median of `blocks` blocks.  A side leg of the default run (world size 1): the per-GPU rates of those
    configurations next to the headline in ONE driver-run line; their parity is the GPU test suite's
    (test_config4_share_* / test_config5_*).
    policy=True: the same launch with actions = NULL -- the in-kernel policy stream (oracle/philox.py random_actions_np: one
    Philox call per agent quad per four steps) stands in for the caller-side `a = policy(s)`; nothing else changes (the
    trajectory is still recorded), and the roofline contract is the same 5 + 18/A bytes per agent-step -- the action byte is
    then produced on the device instead of read from HBM."""
    import ctypes
    import torch
    from gym_mapf_amd import _native as nat
    from gym_mapf_amd.envs.vec_env import OptimizationCriteria, VecMapfEnv
    cfg = CONFIGS[name]
    A, E = cfg['agents'], n_envs
    grid, _, nbr, start, goal = workload_tables(cfg, E, 0)
    env = VecMapfEnv(grid, A, None, None, cfg['fail_prob'], R_CLASH, R_GOAL, R_LIVING, OptimizationCriteria.Makespan, seed=SEED,
                     device=torch.cuda.current_device(), device_arrays=True, start_local=start, goal_local=goal)
    actions = None if policy else env.fill_random_actions(0, 2 * T)
    rec = {'local': env._empty((T, E, A), np.uint16), 'reward': env._empty((T, E), np.float64), 'prob': env._empty((T, E), np.float64),
           'done': env._empty((T, E), np.uint8), 'collision': env._empty((T, E), np.uint8)}
    acc = {'returns': torch.zeros(E, dtype=torch.float64, device='cuda'),
           'episodes': torch.zeros(E, dtype=torch.int32, device='cuda').view(torch.uint32),
           'collisions': torch.zeros(E, dtype=torch.int32, device='cuda').view(torch.uint32)}
    ios = [nat.MapfRolloutIO(struct_size=ctypes.sizeof(nat.MapfRolloutIO), n_steps=T, step_flags=nat.MAPF_STEP_AUTO_RESET, accumulate=1,
                             actions=None if policy else actions[slot * T].data_ptr(), out_returns=acc['returns'].data_ptr(),
                             out_episodes=acc['episodes'].data_ptr(), out_collisions=acc['collisions'].data_ptr(),
                             rec_local=rec['local'].data_ptr(), rec_reward=rec['reward'].data_ptr(), rec_done=rec['done'].data_ptr(),
                             rec_collision=rec['collision'].data_ptr(), rec_prob=rec['prob'].data_ptr()) for slot in range(2)]
    env.reset()
    k = 0
    t_end = time.perf_counter() + preroll_ms * 1e-3
    while time.perf_counter() < t_end:
        for _ in range(4):
            nat.check(env._lib.mapf_rollout(env._h, ctypes.byref(ios[k % 2])))
            k += 1
        env.sync()
    ms = []
    for _ in range(blocks):
        env.sync()
        env.timer_begin()
        for _ in range(n_launch):
            nat.check(env._lib.mapf_rollout(env._h, ctypes.byref(ios[k % 2])))
            k += 1
        ms.append(env.timer_end() / n_launch)
        env.sync()
    launch_ms = sorted(ms)[(len(ms) - 1) // 2]
    kernel = env.last_kernel('rollout')
    env.close()
    launch_bytes = float(T) * E * A * bytes_per_agent_step(A)
    traffic = measured_traffic(kernel, E, A, T)
    return {"workload": "%s%s: %s map, %d agents, slip=%g, %d envs%s" % (cfg['baseline'], "" if E == cfg['envs'] else "'s share of one GPU", cfg['map'], A,
                                                                   cfg['fail_prob'], E, ", actions from the in-kernel policy stream (actions = NULL)" if policy else ""),
            "value": float(T) * E * A / (launch_ms * 1e-3), "unit": "agent-steps/s", "ms_per_launch_hip_events": launch_ms,
            "launches": n_launch, "blocks": blocks, "kernel": kernel,
            "roofline": {"bound": "hbm", "achieved": launch_bytes / (launch_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": launch_bytes / (launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": traffic,
                         "bytes_per_launch": launch_bytes}}


def baseline_config_leg(name, rank, world, dist, coll_dev, device, T=256, n_launch=10, blocks=3, preroll_ms=60.0, check_steps=6):
    """One of the BASELINE configurations that are DEFINED over several GPUs -- configs[3] (262144 envs of 8 agents) or
    configs[4] (131072 envs of 32 agents) -- sharded over the ranks of this run: rank r steps
    its block-aligned shard of the global env ids (sharding.split_evenly, the granule of the packed kernels) with the same
    fused launch as the headline leg (T env-steps per launch, every env-step's outputs recorded, actions streamed from a
    two-slot ring); `blocks` timed blocks of `n_launch` launches, each between barrier + synchronize on both sides, MAX over
    ranks; the per-env returns are gathered (the path's one collective); rank 0's shard is checked against the C oracle
    first.  Every rank calls this; the dict is meaningful on rank 0."""
    import ctypes
    import torch
    from gym_mapf_amd import _native as nat
    from gym_mapf_amd import sharding
    from gym_mapf_amd.envs.vec_env import OptimizationCriteria, VecMapfEnv
    cfg = CONFIGS[name]
    A = cfg['agents']
    counts = [sharding.split_evenly(cfg['envs'], r, world, granule=SHARD_GRANULE)[1] for r in range(world)]
    offset, E = sharding.split_evenly(cfg['envs'], rank, world, granule=SHARD_GRANULE)
    grid, _, nbr, start, goal = workload_tables(cfg, E, offset)
    env = VecMapfEnv(grid, A, None, None, cfg['fail_prob'], R_CLASH, R_GOAL, R_LIVING, OptimizationCriteria.Makespan, seed=SEED,
                     env_id_offset=offset, device=device, device_arrays=True, start_local=start, goal_local=goal)
    actions = env.fill_random_actions(0, 2 * T)
    rec = {'local': env._empty((T, E, A), np.uint16), 'reward': env._empty((T, E), np.float64), 'prob': env._empty((T, E), np.float64),
           'done': env._empty((T, E), np.uint8), 'collision': env._empty((T, E), np.uint8)}
    acc = {'returns': torch.zeros(E, dtype=torch.float64, device='cuda'),
           'episodes': torch.zeros(E, dtype=torch.int32, device='cuda').view(torch.uint32),
           'collisions': torch.zeros(E, dtype=torch.int32, device='cuda').view(torch.uint32)}

    def barrier():
        torch.cuda.synchronize()
        env.sync()
        if dist is not None:
            dist.barrier()

    parity = None
    if rank == 0:        # rank 0's shard, every env, the first steps of the fused kernel this leg times
        sys.path.insert(0, os.path.join(ROOT, 'oracle'))
        import c_oracle
        import mapf_oracle as mo
        co = c_oracle.COracle(nbr, A, start, goal, cfg['fail_prob'], R_CLASH, R_GOAL, R_LIVING, mo.MAKESPAN, seed=SEED, env_id_offset=offset)
        res = env.rollout(check_steps, actions=actions[:check_steps], auto_reset=True, record=True)
        env.sync()
        ok = True
        for t in range(check_steps):
            ref = co.step(actions[t].cpu().numpy(), auto_reset=True)
            ok &= bool(np.array_equal(res['local'][t].cpu().numpy(), ref['local']))
            ok &= bool(np.array_equal(res['reward'][t].cpu().numpy().view(np.uint64), ref['reward'].view(np.uint64)))
            ok &= bool(np.array_equal(res['prob'][t].cpu().numpy().view(np.uint64), ref['prob'].view(np.uint64)))
            ok &= bool(np.array_equal(res['done'][t].cpu().numpy(), ref['done']))
            ok &= bool(np.array_equal(res['collision'][t].cpu().numpy(), ref['collision']))
        if not ok:
            raise SystemExit('PARITY FAILURE (%s shard of rank 0): HIP path differs from the oracle' % name)
        parity = {"checked_env_steps": check_steps * E, "bit_exact": True, "against": "oracle/mapf_oracle.c",
                  "covers": "a %d-step fused rollout (recorded trajectory), every env of rank 0's shard" % check_steps}
        del res
        env.reset()
        env.set_state(None, t=0)
    ios = [nat.MapfRolloutIO(struct_size=ctypes.sizeof(nat.MapfRolloutIO), n_steps=T, step_flags=nat.MAPF_STEP_AUTO_RESET, accumulate=1,
                             actions=actions[slot * T].data_ptr(), out_returns=acc['returns'].data_ptr(),
                             out_episodes=acc['episodes'].data_ptr(), out_collisions=acc['collisions'].data_ptr(),
                             rec_local=rec['local'].data_ptr(), rec_reward=rec['reward'].data_ptr(), rec_done=rec['done'].data_ptr(),
                             rec_collision=rec['collision'].data_ptr(), rec_prob=rec['prob'].data_ptr()) for slot in range(2)]
    k = 0
    barrier()
    t_end = time.perf_counter() + preroll_ms * 1e-3
    while time.perf_counter() < t_end:
        for _ in range(4):
            nat.check(env._lib.mapf_rollout(env._h, ctypes.byref(ios[k % 2])))
            k += 1
        env.sync()
    timed = []
    for _ in range(blocks):
        barrier()
        env.timer_begin()
        t0 = time.perf_counter()
        for _ in range(n_launch):
            nat.check(env._lib.mapf_rollout(env._h, ctypes.byref(ios[k % 2])))
            k += 1
        gpu_ms = env.timer_end()
        torch.cuda.synchronize()
        env.sync()
        wall = time.perf_counter() - t0
        barrier()
        if dist is not None:
            tmax = torch.tensor([wall], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            wall = float(tmax.item())
        timed.append((wall, gpu_ms))
    order = sorted(range(len(timed)), key=lambda i: timed[i][0])
    wall, gpu_ms = timed[order[(len(timed) - 1) // 2]]
    kernel = env.last_kernel('rollout')
    packed = 1 if kernel.startswith('lq_rollout_kernel') else 0
    gathered_n = E
    if dist is not None:
        env.sync()
        gathered = sharding.gather_returns(acc['returns'] if coll_dev == 'cuda' else acc['returns'].cpu(), counts=counts)
        torch.cuda.synchronize()
        gathered_n = int(gathered.numel())
        flag = torch.tensor([packed], dtype=torch.int32, device=coll_dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        packed = int(flag.item())
    env.close()
    del rec, acc, actions
    torch.cuda.empty_cache()
    launch_ms = gpu_ms / n_launch
    launch_bytes = float(T) * E * A * bytes_per_agent_step(A)
    return {"workload": "%s: %s map, %d agents, slip=%g, %d envs in total over %d GPU(s)" % (cfg['baseline'], cfg['map'], A, cfg['fail_prob'], cfg['envs'], world),
            "value": float(n_launch) * T * cfg['envs'] * A / wall, "unit": "agent-steps/s", "scaling": "strong",
            "ms_per_step": wall * 1e3 / n_launch, "steps": n_launch, "blocks": blocks, "env_steps_per_step": T, "envs_total": cfg['envs'],
            "shards": counts, "kernel": kernel, "packed_kernel_on_every_rank": bool(packed),
            "gather": {"elements": gathered_n, "collective": None if dist is None else ("all_gather_into_tensor" if coll_dev == 'cuda' else "all_gather")},
            "roofline": {"bound": "hbm", "of": "rank 0's shard (%d envs), HIP events" % E, "achieved": launch_bytes / (launch_ms * 1e-3) / 1e9,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": launch_bytes / (launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "bytes_per_launch": launch_bytes, "ms_per_launch_hip_events": launch_ms,
                         "traffic": measured_traffic(kernel, E, A, T)},
            "parity": parity}


def transitions_rate(n_agents, n_queries, reps=10, blocks=3, compact=False, seed=0, preroll_ms=150.0):
    """`env.P[s][a]` (reference mapf_env.py:448-483) as one mapf_transitions launch over `n_queries` random
    (state, joint action) queries on room-32-32-4: distinct random cells per query, uniform joint actions, outputs reserved
    once.  The kernel reads 3A bytes per query and WRITES every branch of the joint slip distribution -- next cells u16[A],
    prob f64, reward f64, done u8, collision u8 = 2A + 18 bytes per branch, nothing re-read -- so its roofline is HBM write
    bandwidth: achieved = branches x (2A + 18) / HIP-event time per launch (median of `blocks` blocks of `reps` launches).
    compact=True: the packed output mode (rows of all queries back to back behind an exclusive scan of the branch counts).
    `preroll_ms` of untimed launches come first, as for the headline (--preroll-ms): the queries are drawn on the host while the
    GPU idles, and 20 launches of a quarter millisecond are over before its clocks have come back -- without them the same
    binary measured 100 or 132 G branches/s from one process to the next (profiles/r05_transitions_magic_digits_ab.txt)."""
    import torch
    from gym_mapf_amd.envs import map_name_to_files
    from gym_mapf_amd.envs.grid import MapfGrid
    from gym_mapf_amd.envs.utils import parse_map_file, parse_scen_file
    from gym_mapf_amd.envs.vec_env import OptimizationCriteria, VecMapfEnv
    A, N = int(n_agents), int(n_queries)
    grid = MapfGrid(parse_map_file(map_name_to_files('room-32-32-4', 6)[0]))
    starts, goals = parse_scen_file(map_name_to_files('room-32-32-4', 6)[1], A)
    env = VecMapfEnv(grid, A, starts, goals, 0.2, R_CLASH, R_GOAL, R_LIVING, OptimizationCriteria.Makespan, n_envs=1,
                     device=torch.cuda.current_device(), device_arrays=True)
    rs = np.random.RandomState(seed)
    V = env.n_cells
    local = rs.randint(0, V, size=(N, A))
    while True:                                                   # rows with a repeated cell are redrawn until none is left
        srt = np.sort(local, axis=1)
        bad = np.nonzero((srt[:, 1:] == srt[:, :-1]).any(axis=1))[0]
        if bad.size == 0:
            break
        local[bad] = rs.randint(0, V, size=(bad.size, A))
    local = local.astype(np.uint16)
    acts = rs.randint(0, 5, size=(N, A)).astype(np.uint8)
    lt = torch.from_numpy(local.view(np.int16)).cuda().view(torch.uint16)
    at = torch.from_numpy(acts).cuda()
    M = 3 ** A
    if compact:
        res = env.transitions_compact(lt, at)
        call = lambda: env.transitions_compact(lt, at, out=res)
    else:
        res = env.transitions(lt, at, max_branches=M)
        call = lambda: env.transitions(lt, at, max_branches=M, out=res)
    env.sync()
    branches = int(res['count'].to(torch.int64).sum().item())
    t_end = time.perf_counter() + preroll_ms * 1e-3
    while time.perf_counter() < t_end:
        for _ in range(4):
            call()
        env.sync()
    ms = []
    for _ in range(blocks):
        env.sync()
        env.timer_begin()
        for _ in range(reps):
            call()
        ms.append(env.timer_end() / reps)
    launch_ms = sorted(ms)[(len(ms) - 1) // 2]
    kernel = env.last_kernel('transitions')
    env.close()
    nbytes = float(branches) * (2 * A + 18)
    return {"workload": "room-32-32-4, %d agents, %d random (state, joint action) queries, slip 0.2, %s output rows"
                        % (A, N, 'compacted' if compact else '3^A reserved'),
            "value": branches / (launch_ms * 1e-3), "unit": "branches/s", "branches": branches, "queries": N,
            "ms_per_launch_hip_events": launch_ms, "launches": reps, "blocks": blocks, "kernel": kernel,
            "roofline": {"bound": "hbm", "achieved": nbytes / (launch_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": nbytes / (launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "bytes_per_launch": nbytes,
                         "bytes_per_branch": 2 * A + 18, "traffic": None}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200, help='timed passes (one pass = one fused launch of --rollout-steps env-steps)')
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--config', default='c3', choices=sorted(CONFIGS))
    ap.add_argument('--envs', type=int, default=None, help='override the config: envs per GPU (weak scaling)')
    ap.add_argument('--rollout-steps', type=int, default=256, help='T: env-steps fused per launch')
    ap.add_argument('--preroll-ms', type=float, default=150.0,
                    help='untimed launches before the warm-up steps, so that the timed region sees sustained clocks')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-side-legs', action='store_true', help='skip single_step_launches and scalar_env (profiling runs)')
    ap.add_argument('--no-scalar-env', action='store_true', help='skip the scalar_env leg (its ~100k one-env launches swamp a profile)')
    ap.add_argument('--no-per-gpu-shapes', action='store_true',
                    help='skip the per_gpu_shapes leg (profiling runs: it launches the headline kernel at another batch size)')
    ap.add_argument('--no-baseline-configs', action='store_true',
                    help='multi-rank runs: skip the baseline_configs leg (configs[3] and configs[4] sharded over the ranks)')
    ap.add_argument('--baseline-config-steps', type=int, default=10, help='launches per timed block of the baseline_configs leg')
    ap.add_argument('--policy-actions', action='store_true',
                    help='the headline leg with actions = NULL (in-kernel policy stream) -- the counter passes of the policy kernel '
                         '(tools/refresh_profiles.sh c3p); the default line carries that launch as the policy_rollout side leg')
    ap.add_argument('--no-transitions', action='store_true', help='skip the transitions leg (env.P enumeration)')
    ap.add_argument('--no-policy-rollout', action='store_true', help='skip the policy_rollout leg (the same batch with actions = NULL)')
    ap.add_argument('--kernel', default='auto', choices=['auto', 'thread_per_env', 'lane_group'])
    ap.add_argument('--dist-backend', default='nccl', choices=['nccl', 'gloo'],
                    help="'gloo' + --share-device rehearses the N > 1 path on a one-GPU box")
    ap.add_argument('--share-device', action='store_true', help='every rank uses cuda:0 (rehearsal only)')
    ap.add_argument('--force-dist', action='store_true',
                    help='initialise torch.distributed and run the collective legs (MAX all-reduce, gather of the returns) even '
                         'with --gpus 1: with --dist-backend nccl this is RCCL at world size 1 -- what a one-GPU box can verify')
    ap.add_argument('--repeats', type=int, default=5, help='timed blocks of --steps steps each; value = their median')
    ap.add_argument('--rank-timeout', type=float, default=900.0,
                    help='self-launched ranks (--gpus N without a launcher) are all terminated after this many seconds')
    ap.add_argument('--cpu-worker', nargs=4, metavar=('KIND', 'CONFIG', 'SECONDS', 'INDEX'), default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.cpu_worker:
        cpu_worker(args.cpu_worker[0], args.cpu_worker[1], float(args.cpu_worker[2]), int(args.cpu_worker[3]))
        return
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        spawn_ranks(args.gpus, args.rank_timeout)
    # the whole-host CPU figure runs FIRST: its workers are child programs, which a process may only start before it touches the GPU
    all_cores = None
    if args.gpus == 1 and not args.force_dist and not args.no_cpu_baseline and int(os.environ.get('WORLD_SIZE', '1')) == 1:
        all_cores = cpu_all_cores(args.config)

    # stdout carries ONE JSON line: whatever libraries print while they initialise (gloo's rank banner, ...) goes to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    if args.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if world == 1:   # --force-dist without a launcher
            os.environ.setdefault('MASTER_PORT', str(29500 + os.getpid() % 2000))
            os.environ.setdefault('RANK', '0')
            os.environ.setdefault('WORLD_SIZE', '1')
        if args.dist_backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group('gloo')
    coll_dev = 'cuda' if args.dist_backend == 'nccl' else 'cpu'

    from gym_mapf_amd.envs.vec_env import OptimizationCriteria, VecMapfEnv
    from gym_mapf_amd import sharding
    cfg = CONFIGS[args.config]
    A, K, W = cfg['agents'], max(1, args.steps), max(0, args.warmup)
    scaling = cfg['scaling']
    if args.envs is not None:
        scaling = 'weak'
        offset, E = sharding.shard_offset(args.envs, rank), args.envs
        total_envs = args.envs * world
    elif scaling == 'weak':
        offset, E = sharding.shard_offset(cfg['envs'], rank), cfg['envs']
        total_envs = cfg['envs'] * world
    else:
        offset, E = sharding.split_evenly(cfg['envs'], rank, world, granule=SHARD_GRANULE)
        total_envs = cfg['envs']
    grid, _, nbr, start, goal = workload_tables(cfg, E, offset)
    env = VecMapfEnv(grid, A, None, None, cfg['fail_prob'], R_CLASH, R_GOAL, R_LIVING, OptimizationCriteria.Makespan,
                     seed=SEED, env_id_offset=offset, device=local_rank, device_arrays=True,
                     start_local=start, goal_local=goal, kernel=args.kernel)
    T = max(1, args.rollout_steps)
    n_slots = 2
    ring = n_slots * T
    actions = env.fill_random_actions(0, ring)                      # [ring, E, A] u8, resident in HBM
    env.sync()

    def barrier():
        torch.cuda.synchronize()
        env.sync()
        if dist is not None:
            dist.barrier()

    # ---- parity: first steps of this run vs the C oracle, every env of this rank
    parity = None
    if rank == 0:
        sys.path.insert(0, os.path.join(ROOT, 'oracle'))
        import c_oracle
        import mapf_oracle as mo
        co = c_oracle.COracle(nbr, A, start, goal, cfg['fail_prob'], R_CLASH, R_GOAL, R_LIVING, mo.MAKESPAN,
                              seed=SEED, env_id_offset=offset)
        n_chk, ok = 6, True
        for t in range(n_chk):
            local, reward, done, info = env.step(actions[t % ring], auto_reset=True)
            env.sync()
            ref = co.step(actions[t % ring].cpu().numpy(), auto_reset=True)
            ok &= bool(np.array_equal(local.cpu().numpy(), ref['local']))
            ok &= bool(np.array_equal(reward.cpu().numpy().view(np.uint64), ref['reward'].view(np.uint64)))
            ok &= bool(np.array_equal(info['prob'].cpu().numpy().view(np.uint64), ref['prob'].view(np.uint64)))
            ok &= bool(np.array_equal(done.cpu().numpy(), ref['done']))
            ok &= bool(np.array_equal(info['collision'].cpu().numpy(), ref['collision']))
        # ... and the same for the fused rollout kernel the headline leg times: recorded trajectory of its first steps
        env.reset()
        env.set_state(None, t=0)
        co.reset()
        co.t = 0
        n_ro = min(8, ring)
        res = env.rollout(n_ro, actions=None if args.policy_actions else actions[:n_ro], auto_reset=True, record=True)
        env.sync()
        if args.policy_actions:
            import philox
        for t in range(n_ro):
            ref = co.step(philox.random_actions_np(SEED, offset + np.arange(E), t, A) if args.policy_actions else actions[t].cpu().numpy(), auto_reset=True)
            ok &= bool(np.array_equal(res['local'][t].cpu().numpy(), ref['local']))
            ok &= bool(np.array_equal(res['reward'][t].cpu().numpy().view(np.uint64), ref['reward'].view(np.uint64)))
            ok &= bool(np.array_equal(res['prob'][t].cpu().numpy().view(np.uint64), ref['prob'].view(np.uint64)))
            ok &= bool(np.array_equal(res['done'][t].cpu().numpy(), ref['done']))
            ok &= bool(np.array_equal(res['collision'][t].cpu().numpy(), ref['collision']))
        parity = {"checked_env_steps": (n_chk + n_ro) * E, "bit_exact": ok, "against": "oracle/mapf_oracle.c",
                  "covers": "%d single-step launches + a %d-step fused rollout (recorded trajectory), every env of rank 0"
                            % (n_chk, n_ro)}
        if not ok:
            raise SystemExit('PARITY FAILURE: HIP path differs from the oracle')
        del res
        env.reset()
        env.set_state(None, t=0)

    import ctypes
    from gym_mapf_amd import _native as nat
    bpas = bytes_per_agent_step(A)

    def block(enqueue, first, n):
        """One timed block: barrier + sync, n enqueues bracketed by HIP events on the env's stream, sync + barrier;
        returns (max-over-ranks wall seconds, HIP-event milliseconds of this rank)."""
        barrier()
        env.timer_begin()
        t0 = time.perf_counter()
        for k in range(n):
            enqueue(first + k)
        gpu_ms = env.timer_end()
        torch.cuda.synchronize()
        env.sync()
        wall = time.perf_counter() - t0                   # this rank's n steps, device idle again; MAX over ranks below
        barrier()
        if dist is not None:
            tmax = torch.tensor([wall], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            wall = float(tmax.item())
        return wall, gpu_ms

    def timed(enqueue, n_warm, n_timed, repeats, cold=0):
        """`repeats` timed blocks of exactly n_timed enqueues each (see block()); returns (list of (wall s, gpu ms), the
        cold block or None).  cold > 0: before anything else the device idles for a second and the first `cold`
        enqueues are timed as a block of their own -- what a caller sees who does not keep the device busy.  Then the
        same launches run untimed for --preroll-ms (after an idle period the device needs ~25 ms of work to reach the
        clocks it sustains, profiles/r02_launch_series.txt), then the n_warm warm-up steps, then the blocks."""
        cold_block = None
        k = 0
        if cold:
            env.sync()
            time.sleep(1.0)
            cold_block = block(enqueue, 0, cold)
            k = cold
        t_end = time.perf_counter() + args.preroll_ms * 1e-3
        while time.perf_counter() < t_end:
            for _ in range(8):
                enqueue(k)
                k += 1
            env.sync()
        for _ in range(n_warm):
            enqueue(k)
            k += 1
        blocks = []
        for _ in range(max(1, repeats)):
            blocks.append(block(enqueue, k, n_timed))
            k += n_timed
        return blocks, cold_block

    def median_block(blocks):
        order = sorted(range(len(blocks)), key=lambda i: blocks[i][0])
        return blocks[order[(len(blocks) - 1) // 2]]        # a measured block (lower median), not an average of two

    # ---- headline leg: K passes, each ONE fused mapf_rollout launch of T env-steps.  Every env-step's outputs
    # (next cells, reward, done, collision, prob) are written to HBM, actions are streamed from the ring.
    rec = {'local': env._empty((T, E, A), np.uint16), 'reward': env._empty((T, E), np.float64),
           'prob': env._empty((T, E), np.float64), 'done': env._empty((T, E), np.uint8),
           'collision': env._empty((T, E), np.uint8)}
    acc = {'returns': torch.zeros(E, dtype=torch.float64, device='cuda'),
           'episodes': torch.zeros(E, dtype=torch.int32, device='cuda').view(torch.uint32),
           'collisions': torch.zeros(E, dtype=torch.int32, device='cuda').view(torch.uint32)}

    def rollout_io(slot):
        return nat.MapfRolloutIO(
            struct_size=ctypes.sizeof(nat.MapfRolloutIO), n_steps=T, step_flags=nat.MAPF_STEP_AUTO_RESET,
            accumulate=1, actions=None if args.policy_actions else actions[slot * T].data_ptr(), out_returns=acc['returns'].data_ptr(),
            out_episodes=acc['episodes'].data_ptr(), out_collisions=acc['collisions'].data_ptr(),
            rec_local=rec['local'].data_ptr(), rec_reward=rec['reward'].data_ptr(), rec_done=rec['done'].data_ptr(),
            rec_collision=rec['collision'].data_ptr(), rec_prob=rec['prob'].data_ptr())

    ios = [rollout_io(slot) for slot in range(n_slots)]

    def enqueue_rollout(k):
        nat.check(env._lib.mapf_rollout(env._h, ctypes.byref(ios[k % n_slots])))

    env.reset()
    n_cold = 25
    ro_blocks, ro_cold = timed(enqueue_rollout, W, K, args.repeats, cold=n_cold)
    wall, gpu_ms = median_block(ro_blocks)
    rollout_kernel = env.last_kernel('rollout')
    agent_steps = float(K) * T * total_envs * A
    value = agent_steps / wall
    ro_launch_ms = gpu_ms / K
    ro_bytes = float(T) * E * A * bpas                              # algorithmic bytes of one launch of this rank
    ro_achieved = ro_bytes / (ro_launch_ms * 1e-3) / 1e9
    gather_info = None
    if dist is not None:
        # the one collective of the path: gather per-env episode returns
        env.sync()
        counts = [sharding.split_evenly(cfg['envs'], r, world, granule=SHARD_GRANULE)[1] for r in range(world)] if scaling == 'strong' else [E] * world
        gathered = sharding.gather_returns(acc['returns'] if coll_dev == 'cuda' else acc['returns'].cpu(), counts=counts)
        torch.cuda.synchronize()
        assert gathered.numel() == total_envs, (gathered.numel(), total_envs)
        gather_info = {"backend": args.dist_backend, "elements": int(gathered.numel()), "shards": counts,
                       "collective": "all_gather_into_tensor" if coll_dev == 'cuda' else "all_gather"}

    # ---- the BASELINE configurations that are defined over several GPUs, sharded over THIS run's ranks (multi-rank runs:
    # what a driver scaling run must report for configs[3] / configs[4]; the headline above stays configs[2] per GPU)
    baseline_legs = None
    if dist is not None and not args.no_baseline_configs and args.config == 'c3':
        baseline_legs = {}
        for leg_name, key in (('c4', 'configs[3]'), ('c5', 'configs[4]')):
            baseline_legs[key] = baseline_config_leg(leg_name, rank, world, dist, coll_dev, local_rank, T=T,
                                                     n_launch=args.baseline_config_steps)

    # ---- second leg: env-steps as single-step mapf_step launches (one kernel launch per env-step)
    single = None
    if not args.no_side_legs:
        # recorded once, replayed: 256 mapf_step nodes; the step writes its cells once (the next observation is the
        # handle's state view) and every other output to HBM
        n_nodes = min(256, ring)                                     # (a replay boundary costs ~5 us: hipGraphLaunch + the advance node)
        out = None
        env.reset()
        env.graph_begin()
        for r in range(n_nodes):
            call, out = env.prepare_step(actions[r], auto_reset=True, out=out, write_local=False)
            call()
        graph = env.graph_end()
        K1 = 8                                                       # replays per timed block = 2048 env-steps
        g_blocks, _ = timed(lambda k: graph.launch(1), 4, K1, min(args.repeats, 3))
        wall1, gpu_ms1 = median_block(g_blocks)
        step_kernel = env.last_kernel('step')
        n_steps1 = K1 * n_nodes
        step_ms = gpu_ms1 / n_steps1
        graph.close()
        # the same calls issued one by one from the host
        calls = [env.prepare_step(actions[r], auto_reset=True, out=out, write_local=False)[0] for r in range(ring)]
        p_blocks, _ = timed(lambda k: calls[k % ring](), 200, 2000, 1)
        launch_bytes = E * A * bpas
        st_traffic = measured_traffic(step_kernel, E, A, 1)
        single = {"value": float(n_steps1) * total_envs * A / wall1, "unit": "agent-steps/s", "launches": n_steps1,
                  "ms_per_launch": wall1 * 1e3 / n_steps1, "kernel": step_kernel,
                  "how": "%d mapf_step calls recorded into one hipGraph, %d replays per block; out_local = NULL (state view)" % (n_nodes, K1),
                  "roofline": {"bound": "hbm", "achieved": launch_bytes / (step_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": launch_bytes / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                               "traffic": st_traffic,
                               "traffic_frac": (st_traffic / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if st_traffic else None,
                               "bytes_per_launch": launch_bytes, "ms_per_launch_hip_events": step_ms},
                  "plain_launches": {"value": 2000.0 * total_envs * A / p_blocks[0][0], "unit": "agent-steps/s",
                                     "ms_per_launch": p_blocks[0][0] * 1e3 / 2000, "ms_per_launch_hip_events": p_blocks[0][1] / 2000}}

    ro_traffic = measured_traffic(rollout_kernel, E, A, T)          # PMC bytes per launch (profiles/)
    ro_valu = measured_valu(rollout_kernel, E, A, T)                # SQ pass: VALU wave-instructions, busy share
    if rank == 0:
        line = {
            "metric": "agent-steps/sec (batched MapfEnv.step)", "value": value, "unit": "agent-steps/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": wall * 1e3 / K, "higher_is_better": True,
            "scaling": scaling, "vs_baseline": None, "dtype": "u16/f64", "data": "synthetic",
            "config": {"workload": "%s: %s map, %d agents, slip=%g, %d envs %s, Makespan, auto-reset; one step = one fused "
                                   "mapf_rollout launch of %d MapfEnv.step() calls per env, every env-step's next cells / "
                                   "reward / done / collision / prob written to HBM, actions %s"
                                   % (cfg['baseline'], cfg['map'], A, cfg['fail_prob'],
                                      E if scaling == 'weak' else total_envs,
                                      'per GPU' if scaling == 'weak' else 'in total over %d GPU(s)' % world, T,
                                      'from the in-kernel policy stream' if args.policy_actions else 'streamed from HBM'),
                       "name": args.config, "envs_per_gpu": E, "envs_total": total_envs, "n_agents": A,
                       "fail_prob": cfg['fail_prob'], "seed": SEED, "env_steps_per_step": T,
                       "agent_steps_per_step": T * total_envs * A, "action_ring_env_steps": ring,
                       "preroll_ms": args.preroll_ms,
                       "parallelism": "env-sharded x%d" % world},
            "value_hip_events": float(T) * total_envs * A / (ro_launch_ms * 1e-3),
            "value_cold": (float(n_cold) * T * total_envs * A / ro_cold[0]) if ro_cold else None,
            "repeats": {"n": len(ro_blocks), "steps_per_block": K, "value_of": "median block",
                        "values": [agent_steps / b[0] for b in ro_blocks],
                        "ms_per_step": [b[0] * 1e3 / K for b in ro_blocks]},
            "roofline": {"bound": "hbm", "achieved": ro_achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ro_achieved / HBM_PEAK_GBS,
                         "traffic": ro_traffic,
                         "traffic_frac": (ro_traffic / (ro_launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if ro_traffic else None,
                         "valu_frac": (ro_valu[1] / ro_launch_ms) if ro_valu else None,
                         "valu_insts_per_launch": ro_valu[0] if ro_valu else None,
                         "kernel": rollout_kernel, "bytes_per_launch": ro_bytes,
                         "ms_per_launch_hip_events": ro_launch_ms,
                         "note": "achieved = algorithmic bytes (5 + 18/A per agent-step, SURVEY.md 8(d)) per launch / HIP-event "
                                 "time per launch; traffic = PMC-measured HBM bytes per launch (the fused kernel keeps state "
                                 "in registers, so it moves fewer bytes than the per-step contract credits)"},
            "parity": parity,
        }
        if args.share_device and world > 1:
            # every rank ran on cuda:0: a plumbing rehearsal of the N > 1 path, NOT a multi-GPU measurement -- the counters
            # under profiles/ describe one process alone on the device and do not apply
            line["rehearsal"] = True
            line["physical_gpus"] = 1
            for key in ("traffic", "traffic_frac", "valu_frac", "valu_insts_per_launch"):
                line["roofline"][key] = None
        if gather_info is not None:
            line["gather"] = gather_info
        if baseline_legs is not None:
            line["baseline_configs"] = baseline_legs
            if args.share_device and world > 1:
                for leg in baseline_legs.values():
                    leg["rehearsal"] = True
                    leg["roofline"]["traffic"] = None
        if single is not None:
            line["single_step_launches"] = single
        if world == 1 and not args.no_side_legs and not args.no_scalar_env:
            line["scalar_env"] = scalar_env_rate()
        if world == 1 and not args.no_side_legs and not args.no_per_gpu_shapes and args.config == 'c3' and args.envs is None:
            # what each GPU runs of the two BASELINE configurations that are sharded over eight
            line["per_gpu_shapes"] = {"c4_share": per_gpu_shape_rate('c4', CONFIGS['c4']['envs'] // 8),
                                      "c5_share": per_gpu_shape_rate('c5', CONFIGS['c5']['envs'] // 8)}
        if world == 1 and not args.no_side_legs and not args.no_policy_rollout and not args.policy_actions and args.envs is None and args.kernel == 'auto':
            # the same batch with the policy ON the device: mapf_rollout(actions = NULL), trajectory recorded
            line["policy_rollout"] = per_gpu_shape_rate(args.config, E, T=T, n_launch=20, blocks=5, policy=True)
            line["policy_rollout"]["vs_streamed_hip_events"] = line["policy_rollout"]["value"] / line["value_hip_events"]
        if world == 1 and not args.no_side_legs and not args.no_transitions and args.envs is None and args.config == 'c3':
            # env.P[s][a]: every branch of random queries, compacted rows; a write-bound kernel by contract
            line["transitions"] = {"a8_q20000": transitions_rate(8, 20000, compact=True), "a4_q2000000": transitions_rate(4, 2000000, compact=True),
                                   "a8_q20000_reserved_rows": transitions_rate(8, 20000, compact=False)}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cfg)
            if all_cores is not None:
                line["cpu_baseline"]["all_cores"] = all_cores
                line["cpu_baseline"]["gpu_over_all_cores_c_port"] = value / all_cores["c_port"] if all_cores.get("c_port") else None
        sys.stdout.flush()
        os.dup2(json_fd, 1)
        print(json.dumps(line), flush=True)
        os.dup2(2, 1)
    env.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
