"""The N > 1 path on CPU: two gloo ranks shard the env axis, build their slice of the bench workload
and all-gather per-env returns in global-id order.  (The stepping itself is covered on the GPU,
including placement independence of the RNG: test_gpu_parity.py.)"""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT
from gym_mapf_amd import sharding


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, envs_per_rank, out_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    import bench
    offset = sharding.shard_offset(envs_per_rank, rank)
    _, _, _, start, goal = bench.workload_tables(bench.CONFIGS['c3'], envs_per_rank, offset)
    ids = offset + np.arange(envs_per_rank)
    local = torch.from_numpy(ids.astype(np.float64) * 0.5 + start[:, 0])       # a value tied to the global id
    gathered = sharding.gather_returns(local)
    tmax = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)                                 # bench.py's max-over-ranks timing
    np.save(os.path.join(out_dir, 'r%d.npy' % rank), gathered.numpy())
    np.save(os.path.join(out_dir, 's%d.npy' % rank), start)
    assert float(tmax) == world
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_and_gather(tmp_path):
    world, per_rank = 2, 96
    mp.spawn(_worker, args=(world, _free_port(), per_rank, str(tmp_path)), nprocs=world, join=True)
    sys.path.insert(0, ROOT)
    import bench
    _, _, _, start_all, _ = bench.workload_tables(bench.CONFIGS['c3'], world * per_rank, 0)
    got = [np.load(tmp_path / ('r%d.npy' % r)) for r in range(world)]
    assert np.array_equal(got[0], got[1])                                       # every rank sees the same gather
    expect = np.arange(world * per_rank) * 0.5 + start_all[:, 0]
    assert np.array_equal(got[0], expect)                                       # ordered by global env id
    shards = np.concatenate([np.load(tmp_path / ('s%d.npy' % r)) for r in range(world)])
    assert np.array_equal(shards, start_all)                                    # slices tile the global workload


def _uneven_worker(rank, world, port, n_total, out_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    counts = [sharding.split_evenly(n_total, r, world)[1] for r in range(world)]
    offset, count = sharding.split_evenly(n_total, rank, world)
    local = torch.arange(offset, offset + count, dtype=torch.float64) * 0.25
    gathered = sharding.gather_returns(local, counts=counts)
    np.save(os.path.join(out_dir, 'u%d.npy' % rank), gathered.numpy())
    try:
        sharding.gather_returns(local, counts=[count + 1] * world)
    except ValueError:
        np.save(os.path.join(out_dir, 'e%d.npy' % rank), np.ones(1))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_of_unequal_shards(tmp_path):
    """A population that does not divide over the ranks (bench.py --config c4 --gpus 3, ...): shards are padded for
    the collective and cut back, the result is ordered by global env id."""
    world, n_total = 3, 100                                                     # shards of 34, 33, 33
    mp.spawn(_uneven_worker, args=(world, _free_port(), n_total, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / ('u%d.npy' % r)), np.arange(n_total) * 0.25)
        assert (tmp_path / ('e%d.npy' % r)).exists()                            # counts that do not fit are refused


def test_split_evenly_covers_every_env_once():
    for n, world in ((262144, 8), (131072, 8), (10, 4), (7, 8), (0, 3)):
        spans = [sharding.split_evenly(n, r, world) for r in range(world)]
        assert sum(c for _, c in spans) == n
        pos = 0
        for off, cnt in spans:
            assert off == pos
            pos += cnt
    assert sharding.shard_offset(65536, 3) == 196608


def test_split_evenly_keeps_shards_on_block_boundaries():
    """bench.py splits a fixed population in multiples of 1024 envs: with a world size that does not divide it (3, 5, 6,
    7 GPUs) every rank would otherwise get a ragged shard and lose the packed kernels (round-3 advisor finding)."""
    import bench
    for n in (262144, 131072, 100000):                             # (populations below granule x world: the next test)
        for world in (1, 2, 3, 5, 6, 7, 8):
            spans = [sharding.split_evenly(n, r, world, granule=bench.SHARD_GRANULE) for r in range(world)]
            assert sum(c for _, c in spans) == n
            pos = 0
            for r, (off, cnt) in enumerate(spans):
                assert off == pos and off % bench.SHARD_GRANULE == 0
                assert cnt % bench.SHARD_GRANULE == 0 or r == world - 1
                pos += cnt
            if n % bench.SHARD_GRANULE == 0:
                assert max(c for _, c in spans) - min(c for _, c in spans) <= bench.SHARD_GRANULE
    assert [sharding.split_evenly(262144, r, 3, granule=1024)[1] for r in range(3)] == [88064, 87040, 87040]


def test_split_evenly_leaves_no_rank_empty():
    """A population smaller than granule x world (round-4 advisor finding: 4096 envs over 8 ranks gave four ranks nothing,
    n = 1023 gave everything to the last one): the granule shrinks until every rank owns envs."""
    assert [sharding.split_evenly(4096, r, 8, granule=1024)[1] for r in range(8)] == [512] * 8
    assert [sharding.split_evenly(1023, r, 2, granule=1024) for r in range(2)] == [(0, 512), (512, 511)]
    for n, world in ((4096, 8), (1023, 2), (1023, 8), (5000, 3), (9, 8), (8, 8)):
        spans = [sharding.split_evenly(n, r, world, granule=1024) for r in range(world)]
        assert all(c > 0 for _, c in spans) and sum(c for _, c in spans) == n, (n, world, spans)
        assert all(spans[r][0] + spans[r][1] == spans[r + 1][0] for r in range(world - 1))
    assert [sharding.split_evenly(3, r, 8, granule=1024)[1] for r in range(8)] == [1, 1, 1, 0, 0, 0, 0, 0]   # fewer envs than ranks


def test_bench_workloads_depend_on_global_env_ids_only():
    """bench.py --config c4 / c5 split a fixed env population over the ranks (sharding.split_evenly): whatever the
    rank count, env e gets the same scenario -- here for the per-env seeded cells of the synthetic C5 workload and
    the scen-id pattern of C3/C4 -- so results cannot depend on the number of GPUs."""
    import numpy as np
    import bench
    from gym_mapf_amd import sharding
    c5 = bench.CONFIGS['c5']
    whole = bench.workload_tables(c5, 9000, 1000)
    assert whole[3].shape == (9000, 32) and whole[3].max() < len(whole[0].tables()[0])
    assert all(len(set(r.tolist())) == 32 for r in whole[3][::97]) and all(len(set(r.tolist())) == 32 for r in whole[4][::97])
    part = bench.workload_tables(c5, 500, 4000)                 # straddles a 4096-id generator chunk
    assert np.array_equal(part[3], whole[3][3000:3500]) and np.array_equal(part[4], whole[4][3000:3500])
    assert not np.array_equal(whole[3], whole[4])
    c4 = bench.CONFIGS['c4']
    shares = [sharding.split_evenly(c4['envs'], r, 8) for r in range(8)]
    assert shares[3] == (98304, 32768) and sum(c for _, c in shares) == c4['envs']
    a = bench.workload_tables(c4, 12, 98304 + 7)
    b = bench.workload_tables(bench.CONFIGS['c3'], 24, 98304)
    assert np.array_equal(a[3], b[3][7:19]) and np.array_equal(a[4], b[4][7:19])
    assert bench.bytes_per_agent_step(8) == 7.25 and bench.bytes_per_agent_step(32) == 5.5625


def test_self_launched_ranks_share_a_deadline_and_the_first_failure_stops_the_rest(tmp_path):
    """bench.spawn_ranks (`python bench.py --gpus N` without a launcher): every child gets RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_*; a rank still running at the deadline has ALL ranks terminated and the parent exits 124; a failing rank
    stops the others and its exit code is the parent's; all ranks succeeding exits 0."""
    import time
    import pytest
    sys.path.insert(0, ROOT)
    import bench
    stamp = str(tmp_path / 'rank')
    child = ("import os, sys, time; open(%r + os.environ['RANK'], 'w').write(os.environ['WORLD_SIZE'] + ' ' + os.environ['LOCAL_RANK'] + ' ' "
             "+ os.environ['MASTER_ADDR']); mode = sys.argv[1]; r = int(os.environ['RANK']); "
             "time.sleep(60 if mode == 'hang' and r == 1 else 0.2); sys.exit(7 if mode == 'fail' and r == 0 else 0)" % stamp)
    t0 = time.monotonic()
    with pytest.raises(SystemExit) as done:
        bench.spawn_ranks(3, 2.0, [sys.executable, '-c', child, 'hang'])
    assert done.value.code == 124 and time.monotonic() - t0 < 20                      # rank 1 did not get its 60 s
    for r in range(3):
        assert open(stamp + str(r)).read() == '3 %d 127.0.0.1' % r
    with pytest.raises(SystemExit) as done:
        bench.spawn_ranks(2, 30.0, [sys.executable, '-c', child, 'fail'])
    assert done.value.code == 7
    with pytest.raises(SystemExit) as done:
        bench.spawn_ranks(2, 30.0, [sys.executable, '-c', child, 'ok'])
    assert done.value.code == 0
