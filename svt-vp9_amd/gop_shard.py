"""GOP sharding of the hot path over GPUs.

Closed GOPs are independent units of the reference encoder (every intra refresh is a key frame,
Source/Lib/Codec/EbPictureDecisionProcess.c:952, 1596-1603), so GOP g is given to rank g % world and each rank runs
the whole hot path for its GOPs: the data path has NO collective.  The only communication is the timing barrier and
the max-reduce of the elapsed time that bench.py needs, plus (host side) the concatenation of the per-GOP outputs in
presentation order.
"""


def assign_gops(n_gops, world):
    """GOP indices of every rank, round-robin (GOP g -> rank g % world)."""
    return [[g for g in range(n_gops) if g % world == r] for r in range(world)]


def gop_seed(base_seed, gop):
    """Seed of the synthetic content of GOP `gop` (bench.py / tests): independent of how the GOPs are sharded."""
    return base_seed + gop


def merge_in_presentation_order(per_rank_outputs, world):
    """per_rank_outputs[r] = list of (gop, payload) produced by rank r -> payloads ordered by GOP index."""
    flat = [x for r in range(world) for x in per_rank_outputs[r]]
    flat.sort(key=lambda t: t[0])
    return [p for _, p in flat]


def reduce_elapsed(dt, dist=None, device=None):
    """Elapsed time of the slowest rank (bench contract: barrier + synchronize on both sides, MAX over ranks)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(dt)
    import torch
    t = torch.tensor([dt], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def aggregate_rate(units_per_rank_per_step, steps, world, elapsed):
    """Whole-job throughput: every rank processes the same number of units per step (weak scaling)."""
    return units_per_rank_per_step * steps * world / elapsed
