"""Batch sharding over ranks: preimages are independent, so a job of `total` rows is cut into
contiguous index ranges, one per rank, and the only exchange is a gather of the result rows to rank 0.
The Philox streams are keyed by the GLOBAL row index, so the gathered matrix equals the single-rank result."""
import torch
import torch.distributed as dist


def shard_range(rank, world, per_rank):
    """Weak scaling: every rank owns `per_rank` rows; returns (first_index, count) of global row indices."""
    return rank * per_rank, per_rank


def split_rows(total, rank, world):
    """Strong scaling split of `total` rows into `world` nearly equal contiguous ranges."""
    base, rem = divmod(total, world)
    count = base + (1 if rank < rem else 0)
    first = rank * base + min(rank, rem)
    return first, count


def gather_rows(local, dst=0, out=None):
    """Gather equally sized row blocks to `dst` (RCCL on GPUs, gloo on CPU); returns the list on dst, else None."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [local]
    world = dist.get_world_size()
    if dist.get_rank() == dst:
        if out is None:
            out = [torch.empty_like(local) for _ in range(world)]
        dist.gather(local, out, dst=dst)
        return out
    dist.gather(local, None, dst=dst)
    return None
