#!/usr/bin/env python3
"""bench.py -- agent-steps/s of the batched MapfEnv.step() hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[2], SURVEY.md 8(d) "C3"): room-32-32-4, 8 agents, slip 0.2,
65536 envs PER GPU (weak scaling; env e uses scen id {6,12,13,23,24,25}[e mod 6], global env
ids so results do not depend on the rank count), synthetic uniform-random actions resident in
HBM before the timed region, every done env auto-reset as the reference's caller loop does.

A "step" is one MapfEnv.step() of every env of the rank: every output (next cells, reward, done,
collision, prob) is written to HBM.  The headline leg fuses T = 256 steps per mapf_rollout launch
(state stays in registers between steps); K steps = ceil(K/T) launches enqueued back to back on the
env's HIP stream between barrier + synchronize on both sides; rank 0 prints ONE JSON line.  Keys:
  roofline      dominant kernel (lq_rollout_kernel<2,...>) -- algorithmic bytes / HIP-event time per launch; `traffic` =
                HBM bytes per launch from the committed PMC passes, `traffic_gbs` = that figure / the same time
  single_step_launches   the same steps as one mapf_step launch each (launch-latency bound at this size)
  cpu_baseline  the pure-Python restatement of the reference (oracle/, kind "port") timed on
                this box's host cores on a bounded sample (rank 0, N=1 only)
  parity        bit-exact check of the first steps of this very run against the C oracle
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, 'gym-mapf_amd'))

import numpy as np  # noqa: E402

MAP, N_AGENTS, FAIL_PROB = 'room-32-32-4', 8, 0.2
SCEN_IDS = (6, 12, 13, 23, 24, 25)
R_CLASH, R_GOAL, R_LIVING = -1000.0, 100.0, -1.0
SEED = 42
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.3 TB/s achievable)


def bytes_per_agent_step(A):
    """u16 state in + u8 action + u16 state out per agent; f64 reward + f64 prob +
    u8 done + u8 collision per env."""
    return 5.0 + 18.0 / A


def measured_traffic(kernel, steps_per_launch=None):
    """HBM bytes per launch of `kernel`, from the committed rocprofv3 PMC passes of this same command
    (profiles/traffic.json: FETCH_SIZE and WRITE_SIZE collected in separate passes, FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for gfx950).  None when no profile is present."""
    try:
        with open(os.path.join(ROOT, 'profiles', 'traffic.json')) as f:
            entry = json.load(f)['kernels'][kernel]
        if steps_per_launch is not None and entry.get('steps_per_launch') != steps_per_launch:
            return None
        return entry['hbm_bytes_per_launch']
    except (OSError, KeyError, ValueError):
        return None


def workload_tables(n_envs, env_id_offset):
    from gym_mapf_amd.envs import map_name_to_files
    from gym_mapf_amd.envs.grid import MapfGrid
    from gym_mapf_amd.envs.utils import parse_map_file, parse_scen_file
    grid = MapfGrid(parse_map_file(map_name_to_files(MAP, SCEN_IDS[0])[0]))
    _, loc_to_int, nbr = grid.tables()
    per_scen = []
    for sid in SCEN_IDS:
        s, g = parse_scen_file(map_name_to_files(MAP, sid)[1], N_AGENTS)
        per_scen.append(([loc_to_int[l] for l in s], [loc_to_int[l] for l in g]))
    which = (env_id_offset + np.arange(n_envs)) % len(SCEN_IDS)
    start = np.asarray([p[0] for p in per_scen], np.uint16)[which]
    goal = np.asarray([p[1] for p in per_scen], np.uint16)[which]
    return grid, nbr, np.ascontiguousarray(start), np.ascontiguousarray(goal)


def cpu_baseline(budget_s=12.0):
    """Time the CPU restatements of the reference on this box (bounded sample of the same workload).

    value = the pure-Python scalar port (oracle/mapf_oracle.py OracleEnv: one env object, Python
    loops and tuples like the reference), 1 core.  The plain-C port's rate is reported beside it."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import c_oracle
    import mapf_oracle as mo
    import philox
    from gym_mapf_amd.envs import map_name_to_files
    from gym_mapf_amd.envs.utils import parse_map_file, parse_scen_file
    lines = parse_map_file(map_name_to_files(MAP, SCEN_IDS[0])[0])
    starts, goals = parse_scen_file(map_name_to_files(MAP, SCEN_IDS[0])[1], N_AGENTS)
    env = mo.OracleEnv(lines, N_AGENTS, starts, goals, FAIL_PROB, R_CLASH, R_GOAL, R_LIVING, mo.MAKESPAN)
    chunk = 2000
    acts = [philox.random_actions_np(SEED, [0], t, N_AGENTS)[0].tolist() for t in range(chunk)]
    us = [philox.slip_uniforms_np(SEED, [0], t, N_AGENTS)[0].tolist() for t in range(chunk)]
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        for a, u in zip(acts, us):
            _, _, done, _, _, _ = env.step(a, u)
            if done:
                env.reset()
        n += chunk
    py_rate = n * N_AGENTS / (time.perf_counter() - t0)

    E = 16384
    _, nbr, start, goal = workload_tables(E, 0)
    co = c_oracle.COracle(nbr, N_AGENTS, start, goal, FAIL_PROB, R_CLASH, R_GOAL, R_LIVING, mo.MAKESPAN, seed=SEED)
    co.rollout(4)
    t0 = time.perf_counter()
    steps = 0
    while time.perf_counter() - t0 < 3.0:
        co.rollout(16)
        steps += 16
    c_rate = steps * E * N_AGENTS / (time.perf_counter() - t0)
    return {"value": py_rate, "unit": "agent-steps/s", "cores": 1, "kind": "port",
            "sample": "oracle/mapf_oracle.py OracleEnv (pure-Python restatement of MapfEnv.step), room-32-32-4 scen 6, "
                      "8 agents, slip 0.2, one env, %d env-steps with reset on done, 1 core of %d" % (n, os.cpu_count()),
            "c_port_value": c_rate,
            "c_port_sample": "oracle/mapf_oracle.c scalar C, %d envs x %d steps, 1 core" % (E, steps)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=2048)
    ap.add_argument('--warmup', type=int, default=128)
    ap.add_argument('--envs', type=int, default=65536, help='envs per GPU')
    ap.add_argument('--ring', type=int, default=512, help='distinct pre-generated action steps kept in HBM')
    ap.add_argument('--rollout-steps', type=int, default=256, help='T of the fused rollout leg')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--kernel', default='auto', choices=['auto', 'thread_per_env', 'lane_group'])
    ap.add_argument('--dist-backend', default='nccl', choices=['nccl', 'gloo'],
                    help="'gloo' + --share-device rehearses the N > 1 path on a one-GPU box")
    ap.add_argument('--share-device', action='store_true', help='every rank uses cuda:0 (rehearsal only)')
    args = ap.parse_args()

    import torch
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d'
                         % (args.gpus, world, args.gpus))
    if args.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if args.dist_backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group('gloo')
    coll_dev = 'cuda' if args.dist_backend == 'nccl' else 'cpu'

    from gym_mapf_amd.envs.vec_env import OptimizationCriteria, VecMapfEnv
    E, A, K, W = args.envs, N_AGENTS, max(1, args.steps), max(0, args.warmup)
    from gym_mapf_amd import sharding
    offset = sharding.shard_offset(E, rank)
    grid, nbr, start, goal = workload_tables(E, offset)
    env = VecMapfEnv(grid, A, None, None, FAIL_PROB, R_CLASH, R_GOAL, R_LIVING, OptimizationCriteria.Makespan,
                     seed=SEED, env_id_offset=offset, device=local_rank, device_arrays=True,
                     start_local=start, goal_local=goal, kernel=args.kernel)
    ring = max(1, min(args.ring, K + W))
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
        co = c_oracle.COracle(nbr, A, start, goal, FAIL_PROB, R_CLASH, R_GOAL, R_LIVING, mo.MAKESPAN,
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
        res = env.rollout(n_ro, actions=actions[:n_ro], auto_reset=True, record=True)
        env.sync()
        for t in range(n_ro):
            ref = co.step(actions[t].cpu().numpy(), auto_reset=True)
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
        env.reset()
        env.set_state(None, t=0)

    import ctypes
    from gym_mapf_amd import _native as nat
    bpas = bytes_per_agent_step(A)

    def timed(enqueue, n_warm, n_timed):
        """barrier + sync, n_timed enqueues bracketed by HIP events on the env's stream, barrier + sync;
        returns (max-over-ranks wall seconds, HIP-event milliseconds)."""
        for k in range(n_warm):
            enqueue(k)
        barrier()
        env.timer_begin()
        t0 = time.perf_counter()
        for k in range(n_timed):
            enqueue(n_warm + k)
        gpu_ms = env.timer_end()
        barrier()
        wall = time.perf_counter() - t0
        if dist is not None:
            tmax = torch.tensor([wall], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            wall = float(tmax.item())
        return wall, gpu_ms

    # ---- headline leg: K steps as fused mapf_rollout launches of T steps each.  Every step's outputs
    # (next cells, reward, done, collision, prob) are written to HBM, actions are streamed from the ring.
    T = max(1, min(args.rollout_steps, ring, K))
    rec = {'local': env._empty((T, E, A), np.uint16), 'reward': env._empty((T, E), np.float64),
           'prob': env._empty((T, E), np.float64), 'done': env._empty((T, E), np.uint8),
           'collision': env._empty((T, E), np.uint8)}
    acc = {'returns': torch.zeros(E, dtype=torch.float64, device='cuda'),
           'episodes': torch.zeros(E, dtype=torch.int32, device='cuda').view(torch.uint32),
           'collisions': torch.zeros(E, dtype=torch.int32, device='cuda').view(torch.uint32)}
    n_slots = max(1, ring // T)

    def rollout_io(slot, n_steps):
        return nat.MapfRolloutIO(
            struct_size=ctypes.sizeof(nat.MapfRolloutIO), n_steps=n_steps, step_flags=nat.MAPF_STEP_AUTO_RESET,
            accumulate=1, actions=actions[slot * T].data_ptr(), out_returns=acc['returns'].data_ptr(),
            out_episodes=acc['episodes'].data_ptr(), out_collisions=acc['collisions'].data_ptr(),
            rec_local=rec['local'].data_ptr(), rec_reward=rec['reward'].data_ptr(), rec_done=rec['done'].data_ptr(),
            rec_collision=rec['collision'].data_ptr(), rec_prob=rec['prob'].data_ptr())

    ios = [rollout_io(slot, T) for slot in range(n_slots)]
    n_full, tail = divmod(K, T)
    io_tail = rollout_io(n_full % n_slots, tail) if tail else None
    n_launch = n_full + (1 if tail else 0)

    def enqueue_rollout(k):
        io = io_tail if (tail and k == W_launch + n_full) else ios[k % n_slots]
        nat.check(env._lib.mapf_rollout(env._h, ctypes.byref(io)))

    W_launch = max(1, W // T)
    env.reset()
    wall, gpu_ms = timed(enqueue_rollout, W_launch, n_launch)
    agent_steps = float(K) * E * A * world
    value = agent_steps / wall
    ro_launch_ms = gpu_ms / n_launch
    ro_bytes = (float(K) / n_launch) * E * A * bpas                 # algorithmic bytes of an average launch
    ro_achieved = ro_bytes / (ro_launch_ms * 1e-3) / 1e9
    if dist is not None:
        # the one collective of the path: gather per-env episode returns
        env.sync()
        gathered = sharding.gather_returns(acc['returns'] if coll_dev == 'cuda' else acc['returns'].cpu())
        torch.cuda.synchronize()
        assert gathered.numel() == world * E

    # ---- second leg: the same K steps as single-step mapf_step launches (one kernel launch per step)
    out = None
    calls = []
    for r in range(ring):
        call, out = env.prepare_step(actions[r], auto_reset=True, out=out)
        calls.append(call)
    env.reset()
    K1 = min(K, 4000)
    wall1, gpu_ms1 = timed(lambda k: calls[k % ring](), min(W, 200), K1)
    step_ms = gpu_ms1 / K1
    launch_bytes = E * A * bpas
    single = {"value": float(K1) * E * A * world / wall1, "unit": "agent-steps/s", "steps": K1,
              "ms_per_step": wall1 * 1e3 / K1, "kernel": "mapf::lg_step_kernel<4,true,false>",
              "roofline": {"bound": "hbm", "achieved": launch_bytes / (step_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                           "unit": "GB/s", "frac": launch_bytes / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                           "traffic": measured_traffic("lg_step_kernel") if E == 65536 else None,
                           "bytes_per_launch": launch_bytes, "ms_per_launch_hip_events": step_ms}}

    ro_traffic = measured_traffic("rollout_kernel", T) if E == 65536 else None   # PMC bytes per launch (profiles/)
    if rank == 0:
        line = {
            "metric": "agent-steps/sec (batched MapfEnv.step)", "value": value, "unit": "agent-steps/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": wall * 1e3 / K, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u16/f64", "data": "synthetic",
            "config": {"workload": "room-32-32-4 map, 8 agents, slip=0.2, %d envs per GPU (BASELINE configs[2]), Makespan, "
                                   "auto-reset; steps fused %d per mapf_rollout launch, every step's next cells / reward / "
                                   "done / collision / prob written to HBM, actions streamed from HBM" % (E, T),
                       "envs_per_gpu": E, "n_agents": A, "fail_prob": FAIL_PROB, "seed": SEED,
                       "steps_per_launch": T, "launches": n_launch, "action_ring_steps": ring,
                       "parallelism": "env-sharded x%d" % world},
            "roofline": {"bound": "hbm", "achieved": ro_achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ro_achieved / HBM_PEAK_GBS,
                         "traffic": ro_traffic,
                         "traffic_gbs": (ro_traffic / (ro_launch_ms * 1e-3) / 1e9) if ro_traffic else None,
                         "kernel": "mapf::lq_rollout_kernel<2,true,true> (quad-lane layout: 4 agents per lane)", "bytes_per_launch": ro_bytes,
                         "ms_per_launch_hip_events": ro_launch_ms},
            "single_step_launches": single,
            "parity": parity,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line))
    env.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
