"""Env-axis sharding over the GPUs of one node (one process per GPU, torch.distributed).

Envs are independent (the reference has no cross-env state), so rank r of R owns the contiguous
global env ids ``[r * E_per_rank, (r + 1) * E_per_rank)`` and steps them with no communication.
RNG counters use the GLOBAL env id (``env_id_offset`` of the handle), so a result does not depend
on how many ranks there are.  The only collective of the path is the end-of-rollout all-gather of
per-env episode returns (RCCL over xGMI with backend "nccl"; gloo on CPU for tests).
"""


def shard_offset(envs_per_rank, rank):
    """Global id of this rank's first env (weak scaling: every rank owns ``envs_per_rank`` envs)."""
    return int(rank) * int(envs_per_rank)


def split_evenly(n_total, rank, world, granule=1):
    """(offset, count) of rank's slice when ``n_total`` envs are divided over ``world`` ranks
    (strong scaling, e.g. BASELINE config 4: 262144 envs over 8 GPUs).  Shards are whole multiples of
    ``granule`` envs -- the packed kernels take a launch only when the batch fills their blocks (64 to
    1024 envs per block depending on the form), and a ragged shard would fall back to the guarded
    lane-group kernel on EVERY rank of a world size that does not divide the population -- with the
    remaining granules going to the lowest ranks and the last ``n_total % granule`` envs to the last rank.  A population
    of fewer than ``granule * world`` envs would leave ranks EMPTY that way (4096 envs over 8 ranks: four idle GPUs
    behind an ``n_gpus: 8`` line), so the granule is halved until every rank owns at least one."""
    n_total, rank, world, granule = int(n_total), int(rank), int(world), max(1, int(granule))
    while granule > 1 and n_total // granule < world:
        granule //= 2
    blocks, tail = divmod(n_total, granule)
    base, rem = divmod(blocks, world)
    count = (base + (1 if rank < rem else 0)) * granule + (tail if rank == world - 1 else 0)
    offset = (rank * base + min(rank, rem)) * granule
    return offset, count


def gather_returns(local_returns, group=None, counts=None):
    """All-gather per-rank ``returns`` tensors into one tensor ordered by global env id.  ``counts`` = every rank's
    shard size (``split_evenly`` gives the lowest ranks one env more when the population does not divide): shards
    are padded to the largest for the collective and the padding is cut out again.  Works on CUDA tensors with the
    nccl(RCCL) backend and on CPU tensors with gloo."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    n = local_returns.numel()
    counts = [n] * world if counts is None else [int(c) for c in counts]
    if len(counts) != world or counts[dist.get_rank(group)] != n:
        raise ValueError('counts %r do not describe this rank\'s shard of %d' % (counts, n))
    width = max(counts)
    padded = local_returns.contiguous()
    if n < width:
        padded = torch.cat([padded, padded.new_zeros(width - n)])
    if padded.is_cuda:
        out = torch.empty((world * width,), dtype=padded.dtype, device=padded.device)
        dist.all_gather_into_tensor(out, padded, group=group)
        parts = list(out.view(world, width))
    else:
        parts = [torch.empty_like(padded) for _ in range(world)]
        dist.all_gather(parts, padded, group=group)
    if all(c == width for c in counts):
        return torch.cat(parts) if not padded.is_cuda else out
    return torch.cat([p[:c] for p, c in zip(parts, counts)])
