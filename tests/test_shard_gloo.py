"""Multi-process (gloo, world_size 2, CPU) test of the GOP-shard path used by `bench.py --gpus N`.

Each rank runs the oracle ME on the GOPs assigned to it; the gathered per-GOP results must equal a serial run over all
GOPs, in presentation order, and the elapsed-time reduction must return the slowest rank's time.
"""
import importlib.util
import os
import socket
import sys
import zlib

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
N_GOPS = 4


def _load_shard():
    spec = importlib.util.spec_from_file_location("gop_shard", os.path.join(ROOT, "svt-vp9_amd", "gop_shard.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _gop_result(gop):
    """CRC of the oracle's ME results for the B picture of a tiny 3-picture GOP."""
    sys.path.insert(0, HERE)
    import me_configs as MC
    import svt_testlib as T
    S = _load_shard()
    frames = T.gen_clip(136, 72, 3, S.gop_seed(40, gop))
    pics = [T.PaPic(f) for f in frames]
    res, _rcme = T.oracle_me_picture(pics[1], pics[0], pics[2], MC.preset("c1_360p_m9", 2, 1))
    return zlib.crc32(np.ascontiguousarray(res).view(np.uint8).tobytes())


def _worker(rank, world, port, out_path):
    import json
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    S = _load_shard()
    mine = S.assign_gops(N_GOPS, world)[rank]
    out = [(g, _gop_result(g)) for g in mine]
    gathered = [None] * world
    dist.all_gather_object(gathered, out)
    dist.barrier()
    slowest = S.reduce_elapsed(1.0 + rank, dist)          # rank r pretends to have taken 1+r seconds
    if rank == 0:
        with open(out_path, "w") as f:
            json.dump({"merged": S.merge_in_presentation_order(gathered, world), "slowest": slowest,
                       "rate": S.aggregate_rate(16, 3, world, slowest)}, f)
    dist.barrier()
    dist.destroy_process_group()


def test_assign_gops_round_robin():
    S = _load_shard()
    assert S.assign_gops(5, 2) == [[0, 2, 4], [1, 3]]
    assert S.assign_gops(3, 8)[:4] == [[0], [1], [2], []]
    assert S.merge_in_presentation_order([[(0, "a"), (2, "c")], [(1, "b")]], 2) == ["a", "b", "c"]
    assert S.reduce_elapsed(2.5) == 2.5 and S.aggregate_rate(16, 10, 8, 2.0) == 640.0


@pytest.mark.timeout(240)
def test_gop_shard_world2_gloo(tmp_path):
    import json
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "rank0.json")
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker", str(r), "2", str(port), out]) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=200) == 0
    got = json.load(open(out))
    assert got["merged"] == [_gop_result(g) for g in range(N_GOPS)]   # same results, presentation order
    assert got["slowest"] == 2.0                                       # MAX over ranks
    assert got["rate"] == 16 * 3 * 2 / 2.0


if __name__ == "__main__" and len(sys.argv) == 6 and sys.argv[1] == "--worker":
    _worker(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5])


# ---- split-GOP mode: the inter-segment reference hand-off (the one exchange step of the path) ----
N_MINIGOPS = 7


HW, HH = 136, 72      # picture of the hand-off chain: 3 x 2 superblocks, the last row 8 samples high


def _encode_minigop(ref, m):
    """"encode the base picture of mini-GOP m from its reference": the REAL chain of the path on the host (the oracle's inter
    prediction from the padded reference -> transform / quantisation / reconstruction -> skip flags -> masks -> deblocking ->
    padding, tests/encdec_model.py) -- what comes out is the padded reference picture the next mini-GOP needs, every byte of it a
    function of the previous one.  ref / result: the whole padded buffer (three planes) as a uint8 tensor."""
    import ctypes as C
    import numpy as np
    import torch
    sys.path.insert(0, HERE)
    import encdec_model as M
    import svt_testlib as T
    B = T.B
    lib = B.load()
    src_y = T.gen_clip(HW, HH, 1, 300 + m)[0]
    src = (src_y, (src_y[::2, ::2] // 2 + 32).astype(np.uint8), np.full((HH // 2, HW // 2), 128, np.uint8))
    rp = M.RefPic(HW, HH)
    rp.buf[:] = ref.numpy()[:rp.buf.size]
    me = np.zeros((T.n_sb(HW, HH), 85), dtype=B.ME_RESULT_DTYPE)         # zero motion: predict from the co-located reference samples
    me["dist0"] = (np.arange(85) * 37 + m) % 500
    mc = np.zeros((HH // 8, HW // 8), dtype=B.MC_MODE_INFO_DTYPE)
    lf = np.zeros((HH // 8, HW // 8), dtype=B.LF_MODE_INFO_DTYPE)
    assert lib.svt_hip_md_default_picture(me.ctypes.data_as(C.c_void_p), HW, HH, 100, 20, mc.ctypes.data_as(C.c_void_p), lf.ctypes.data_as(C.c_void_p), HW // 8) == 0
    fl = B.EncdecFlags(limit_intra=0, allow_enc_dec_mismatch=0, do_recon=1, apply_loop_filter=1, pad_reference=1)
    thr = B.LfThresh()
    lib.svt_hip_lf_thresh_init(C.byref(thr), 0)
    o = M.oracle_encdec_picture(src, [rp, rp], mc, lf, 160, fl, thr)
    return torch.from_numpy(o["rec"].buf.copy())


def _first_reference():
    import numpy as np
    import torch
    sys.path.insert(0, HERE)
    import encdec_model as M
    import svt_testlib as T
    y = T.gen_clip(HW, HH, 1, 299)[0]
    return torch.from_numpy(M.RefPic(HW, HH).set_padded(y, (y[::2, ::2] // 2 + 32).astype(np.uint8), np.full((HH // 2, HW // 2), 128, np.uint8)).buf.copy())


def _handoff_worker(rank, world, port, out_path):
    import json
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    S = _load_shard()
    ref = torch.zeros_like(_first_reference())            # a padded reference picture buffer (three planes)
    if S.minigop_owner(0, world) == rank:
        ref = _first_reference()                          # the key frame's reconstruction
    log = []
    for m in range(N_MINIGOPS):
        got = S.handoff_reference(dist, ref, m, world, rank)       # no-op for everyone but producer and consumer
        if S.minigop_owner(m, world) == rank:
            log.append((m, bool(got)))
            ref = _encode_minigop(ref, m)
    last = S.minigop_owner(N_MINIGOPS - 1, world)
    out = [None] * world
    dist.all_gather_object(out, (log, int(ref.to(torch.int64).sum()) if rank == last else None))
    if rank == 0:
        with open(out_path, "w") as f:
            json.dump(out, f)
    dist.barrier()
    dist.destroy_process_group()


def test_reference_handoff_chain_gloo_world2(tmp_path):
    import json
    import torch
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "handoff.json")
    mp.spawn(_handoff_worker, args=(2, port, out), nprocs=2, join=True)
    got = json.load(open(out))
    # serial run of the same chain
    ref = _first_reference()
    for m in range(N_MINIGOPS):
        ref = _encode_minigop(ref, m)
    want = int(ref.to(torch.int64).sum())
    logs = {m: rcv for r in range(2) for (m, rcv) in got[r][0]}
    assert sorted(logs) == list(range(N_MINIGOPS))                     # every mini-GOP encoded exactly once
    assert logs[0] is False and all(logs[m] for m in range(1, N_MINIGOPS))   # all but the first waited for their reference
    assert [r for r in range(2) if got[r][1] is not None] == [(N_MINIGOPS - 1) % 2] and got[(N_MINIGOPS - 1) % 2][1] == want


def test_c_side_assignment_functions():
    import ctypes as C
    sys.path.insert(0, HERE)
    import svt_testlib as T
    lib = T.B.load()
    buf = (C.c_int64 * 8)()
    assert lib.svt_hip_gop_assign(C.c_int64(10), 4, 1, buf, 8) == 3 and list(buf[:3]) == [1, 5, 9]
    assert lib.svt_hip_gop_assign(C.c_int64(10), 4, 4, buf, 8) < 0 and lib.svt_hip_gop_assign(C.c_int64(2), 8, 5, buf, 8) == 0
    assert [lib.svt_hip_gop_owner(C.c_int64(g), 8) for g in (0, 7, 8, 19)] == [0, 7, 0, 3]
    assert [lib.svt_hip_minigop_reference_source(C.c_int64(m), 8) for m in (0, 1, 8, 9)] == [-1, 0, 7, 0]
