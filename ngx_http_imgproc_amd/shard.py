"""Request sharding across the GPUs of one node: frames are independent, so frame i goes to
rank i mod world and there is no data-path collective.  torch.distributed is only used by bench.py for the
barrier and the max-over-ranks of the elapsed time (`elapsed_max`)."""


def round_robin(n_items, rank, world):
    """Indices of the items rank `rank` of `world` processes owns: i with i % world == rank."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError("bad rank/world %r/%r" % (rank, world))
    return list(range(rank, n_items, world))


def elapsed_max(seconds, dist=None, device=None):
    """Max over ranks of a local elapsed time (identity when not distributed)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return seconds
    import torch

    t = torch.tensor([seconds], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t[0])


def job_totals(seconds, counts, dist=None, device=None):
    """What bench.py reports for a whole job from each rank's share: the SLOWEST rank's time (MAX over ranks) and the SUM
    over ranks of every entry of `counts` (units processed, bytes moved ...).  -> (seconds, [totals]).  Identity when not
    distributed.  value = totals[0] / seconds is then the whole-job aggregate the driver asks for."""
    counts = [float(c) for c in counts]
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(seconds), counts
    import torch

    tmax = torch.tensor([float(seconds)], dtype=torch.float64, device=device or "cpu")
    tsum = torch.tensor(counts, dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
    return float(tmax[0]), [float(v) for v in tsum]
