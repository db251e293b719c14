"""Channel sharding across the GPUs of one node.

Channels are independent psk_soft_i instances, so the batch shards with no data-path
collective: rank r owns a contiguous block of channels and its own handle.  The only
communication is the barrier around the timed region and the max of the elapsed time."""


def shard_channels(total_channels, world_size, rank):
    """Contiguous block of `total_channels` owned by `rank`: (first, count)."""
    base, extra = divmod(int(total_channels), int(world_size))
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


def max_over_ranks(value, dist=None, device=None):
    """Max of a host scalar over all ranks (identity without a process group)."""
    if dist is None or not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    import torch

    t = torch.tensor([float(value)], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, dist=None, device=None):
    if dist is None or not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    import torch

    t = torch.tensor([float(value)], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
