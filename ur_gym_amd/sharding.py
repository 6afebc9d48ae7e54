"""Multi-GPU layout: environments shard across ranks, one process per GPU, no data-path collective.

Every term of the step is per-environment, so rank r simply owns a contiguous block of the global
environment index space with its own handle, tensors, stream and RNG key.  The only exchange the path may want is
an observation gather for a single learner (RCCL all-gather over xGMI; `gloo` in the CPU tests).
"""
import torch


def shard_range(total_envs: int, rank: int, world_size: int):
    """Contiguous [lo, hi) block of global env ids owned by `rank` (sizes differ by at most one)."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    base, rem = divmod(total_envs, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def rank_seed(seed: int, rank: int) -> int:
    """Per-rank RNG key so that shards draw independent episodes (bench.py uses the same rule)."""
    return int(seed) + 1000 * int(rank)


def gather_observations(local: torch.Tensor, group=None) -> torch.Tensor:
    """All-gather [n_local, D] observation shards into [sum n_local, D] on every rank (equal shard sizes)."""
    import torch.distributed as dist

    world = dist.get_world_size(group)
    out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous(), group=group)
    return out


def max_over_ranks(value: float, device="cpu", group=None) -> float:
    """bench.py timing rule: the slowest rank defines the step time."""
    import torch.distributed as dist

    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
