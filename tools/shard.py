"""Block-range sharding of the corpus across ranks: every 64 KiB block is
independent, so rank r of N owns blocks [r*B, (r+1)*B) and there is NO data-path collective.
The only cross-rank traffic is the measurement itself (barrier + MAX of the elapsed time)."""


def first_block(rank, blocks_per_rank):
    return rank * blocks_per_rank


def block_range(rank, world, blocks_per_rank):
    assert 0 <= rank < world
    lo = first_block(rank, blocks_per_rank)
    return lo, lo + blocks_per_rank


def split_range(k, n, total_blocks):
    """Strong scaling: shard k of n of a FIXED number of blocks (contiguous, sizes differ by at most one)."""
    assert 0 <= k < n
    base, extra = divmod(total_blocks, n)
    lo = k * base + min(k, extra)
    return lo, lo + base + (1 if k < extra else 0)


def max_over_ranks(dist, value, device=None):
    """MAX of a python float over all ranks (identity when dist is None)."""
    if dist is None:
        return value
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(dist, value, device=None):
    if dist is None:
        return value
    import torch
    t = torch.tensor([value], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())
