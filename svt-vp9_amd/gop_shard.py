"""Work distribution of the hot path over GPUs, one process per GPU -- the torch.distributed face of
svt-vp9_amd/host/gop_shard.c, whose functions decide who does what.

Closed GOPs are independent units of the reference encoder (every intra refresh is a key frame,
Source/Lib/Codec/EbPictureDecisionProcess.c:952, 1596-1603): GOP g goes to rank g % world and the data path has NO
collective.  Only when ONE GOP is split across GPUs (latency mode) does a mini-GOP need something from another rank: the
reconstructed, deblocked, padded base-layer picture of the mini-GOP before it -- one point-to-point transfer (RCCL send /
recv over xGMI with the nccl backend; gloo in the CPU tests), `handoff_reference` below.
"""
import ctypes as C
import importlib.util
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_sp = importlib.util.spec_from_file_location("svtvp9_binding", os.path.join(_HERE, "binding.py"))
_B = importlib.util.module_from_spec(_sp)
_sp.loader.exec_module(_B)


def assign_gops(n_gops, world):
    """GOP indices of every rank (svt_hip_gop_assign: GOP g -> rank g % world)."""
    lib = _B.load()
    out = []
    for r in range(world):
        buf = (C.c_int64 * max(1, n_gops))()
        n = lib.svt_hip_gop_assign(C.c_int64(n_gops), world, r, buf, n_gops)
        assert n >= 0
        out.append([int(buf[i]) for i in range(n)])
    return out


def gop_seed(base_seed, gop):
    """Seed of the synthetic content of GOP `gop` (bench.py / tests): independent of how the GOPs are sharded."""
    return base_seed + gop


def merge_in_presentation_order(per_rank_outputs, world):
    """per_rank_outputs[r] = list of (gop, payload) produced by rank r -> payloads ordered by GOP index."""
    flat = [x for r in range(world) for x in per_rank_outputs[r]]
    flat.sort(key=lambda t: t[0])
    return [p for _, p in flat]


def minigop_owner(minigop, world):
    return _B.load().svt_hip_gop_owner(C.c_int64(minigop), world)


def handoff_reference(dist, ref, minigop, world, rank):
    """Split-GOP mode: before mini-GOP `minigop` is encoded, its owner needs the padded reference picture `ref` (a tensor,
    same shape on both sides) that the owner of mini-GOP `minigop - 1` produced.  Point-to-point: the producer sends, the
    consumer receives into `ref`; every other rank does nothing.  Returns True on the rank that received."""
    src = _B.load().svt_hip_minigop_reference_source(C.c_int64(minigop), world)
    dst = minigop_owner(minigop, world)
    if src < 0 or src == dst:
        return False        # first mini-GOP of the GOP, or a single device: the reference is already in place
    if rank == src:
        dist.send(ref, dst)
    elif rank == dst:
        dist.recv(ref, src)
        return True
    return False


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
