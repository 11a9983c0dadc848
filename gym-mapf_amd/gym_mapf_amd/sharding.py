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


def split_evenly(n_total, rank, world):
    """(offset, count) of rank's slice when ``n_total`` envs are divided over ``world`` ranks
    (strong scaling, e.g. BASELINE config 4: 262144 envs over 8 GPUs); the remainder goes to the
    lowest ranks."""
    base, rem = divmod(int(n_total), int(world))
    count = base + (1 if rank < rem else 0)
    offset = rank * base + min(rank, rem)
    return offset, count


def gather_returns(local_returns, group=None):
    """All-gather equal-sized per-rank ``returns`` tensors into one tensor ordered by global env id.
    Works on CUDA tensors with the nccl(RCCL) backend and on CPU tensors with gloo."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    out = torch.empty((world * local_returns.numel(),), dtype=local_returns.dtype, device=local_returns.device)
    if local_returns.is_cuda:
        dist.all_gather_into_tensor(out, local_returns.contiguous(), group=group)
    else:
        parts = [torch.empty_like(local_returns) for _ in range(world)]
        dist.all_gather(parts, local_returns.contiguous(), group=group)
        out = torch.cat(parts)
    return out
