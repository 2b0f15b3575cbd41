"""GPU parity: HIP transform/quant/recon batch (through the C ABI) vs the oracle, bit-exact."""
import ctypes as C

import numpy as np
import pytest

import svt_testlib as T

B = T.B
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    lib = B.load()
    c = C.c_void_p()
    B.check(lib.svt_hip_ctx_create(C.byref(c), 0))
    yield c
    lib.svt_hip_ctx_destroy(c)


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
@pytest.mark.parametrize("do_recon", [True, False])
def test_tq_vs_oracle(ctx, seed, do_recon):
    case = T.make_tq_case(seed, do_recon=do_recon)
    o = T.oracle_tq_batch(case)
    g = T.hip_tq_batch(ctx, case)
    names = ("recon", "qcoeff", "dqcoeff", "eob")
    for n, a, b in zip(names, o, g):
        if n == "recon" and not do_recon:
            continue
        assert np.array_equal(a, b), (n, int(np.sum(a != b)), np.argwhere(a != b)[:8].ravel().tolist())
    assert len(set(o[3].tolist())) > 8  # the case really exercises many eob classes


@pytest.mark.parametrize("width", [264, 260])
def test_tq_rows_not_vector_aligned(ctx, width):
    """plane stride = 8 / 4 mod 16: the rows of a block alternate between the 16-byte vector path and the dword path"""
    case = T.make_tq_case(7, width=width)
    o = T.oracle_tq_batch(case)
    g = T.hip_tq_batch(ctx, case)
    for n, a, b in zip(("recon", "qcoeff", "dqcoeff", "eob"), o, g):
        assert np.array_equal(a, b), (n, int(np.sum(a != b)))


@pytest.mark.parametrize("seed", [11, 12])
def test_tq_extreme_residuals(ctx, seed):
    """+-255 residuals, tiny and huge quantiser steps: saturation / clamp paths."""
    case = T.make_tq_case(seed, extreme=True, qsteps=((4, 4), (1336, 1828), (40, 48)))
    o = T.oracle_tq_batch(case)
    g = T.hip_tq_batch(ctx, case)
    for n, a, b in zip(("recon", "qcoeff", "dqcoeff", "eob"), o, g):
        assert np.array_equal(a, b), (n, int(np.sum(a != b)))


def test_tq_rejects_ungrouped_blocks(ctx):
    case = T.make_tq_case(5)
    case["blocks"] = case["blocks"][::-1].copy()
    with pytest.raises(RuntimeError):
        T.hip_tq_batch(ctx, case)


@pytest.mark.parametrize("seed,extreme", [(1, False), (3, False), (11, True)])
def test_tq_distortion_pairs_vs_oracle(ctx, seed, extreme):
    """T3: coefficient-domain distortion (full_distortion_kernel32bit) fused into the TQ batch, device-pointer ABI."""
    case = T.make_tq_case(seed, extreme=extreme, qsteps=((4, 4), (1336, 1828), (40, 48)) if extreme else ((40, 48), (8, 9), (200, 260)))
    o = T.oracle_tq_batch_dist(case)
    g = T.hip_tq_batch_dist_device(ctx, case)
    for n, a, b in zip(("recon", "qcoeff", "dqcoeff", "eob", "dist"), o, g):
        assert np.array_equal(a, b), (n, int(np.sum(a != b)))
    assert (o[4][:, 0] != o[4][:, 1]).any() and o[4].max() > 0


@pytest.mark.parametrize("seed,extreme,inter_share", [(1, False, 0.5), (2, False, 1.0), (4, False, 0.0), (11, True, 0.5), (12, True, 0.3)])
def test_tq_rd_fused_rate_vs_oracle(ctx, seed, extreme, inter_share):
    """svt_hip_tq_rd_batch_device: distortion + coefficient rate computed behind the quantiser (4x4 blocks walked inside their
    lane with the compiled-in scan orders, bigger blocks from the LDS copy of their coefficients) == the oracle's transform
    stage followed by its coeff_rate_estimate; all four sizes, all transform types (intra blocks keep ADST mixes: row / column
    scans), empty blocks (eob 0), full blocks, CAT6 levels (extreme residuals at tiny steps)."""
    case = T.make_tq_case(seed, extreme=extreme, qsteps=((4, 4), (1336, 1828), (40, 48)) if extreme else ((40, 48), (8, 9), (200, 260)))
    rb = T.add_rate_info(case, seed + 100, inter_share)
    o = T.oracle_tq_rd_batch(case, rb)
    g = T.hip_tq_rd_batch_device(ctx, case)
    for n, a, b in zip(("recon", "qcoeff", "dqcoeff", "eob", "dist", "bits"), o, g):
        assert np.array_equal(a, b), (n, int(np.sum(a != b)), np.argwhere(a != b)[:6].ravel().tolist())
    eob = o[3]
    n2 = 16 << (2 * case["blocks"]["tx_size"].astype(np.int64))
    assert len(set(o[5].tolist())) > 32
    if not extreme:
        assert (eob == 0).any()
    else:
        assert (eob == n2).any() and np.abs(o[1]).max() >= 67     # full blocks and CAT6 tokens occur


def test_tq_rd_many_blocks_persistent_grid(ctx):
    """more groups of blocks than resident workgroups: the persistent, XCD-striped walk covers every block exactly once"""
    case = T.make_tq_case(21, width=1024, height=512)
    rb = T.add_rate_info(case, 5)
    o = T.oracle_tq_rd_batch(case, rb)
    g = T.hip_tq_rd_batch_device(ctx, case)
    for n, a, b in zip(("recon", "qcoeff", "dqcoeff", "eob", "dist", "bits"), o, g):
        assert np.array_equal(a, b), (n, int(np.sum(a != b)))


def test_tq_rd_mode_decision_candidate_form(ctx):
    """The MD call sites of the transform path -- perform_coding_loop in the full loop for Y / Cb / Cr
    (Codec/EbEncDecProcess.c:853, 911, 960) and in the chroma-mode search (:1820, 1873), followed by perform_dist_rate_calc
    (:700-745) -- as ONE svt_hip_tq_rd_batch_device call over (block x candidate): K candidates per transform block share the
    block's src_off and differ in pred_off, do_recon = 0 everywhere and d_recon = NULL (mode decision reconstructs nothing; the
    winner's reconstruction is the encode pass).  1080p block counts (1920 x 1056 = 33 rows of 32 x 32 areas, K = 3 -> 126 171
    blocks), against the oracle's transform stage + coeff_rate_estimate."""
    K = 3
    case = T.make_tq_md_case(31, 1920, 1056, K)
    assert (case["blocks"]["do_recon"] == 0).all() and len(case["blocks"]) > 100000
    b = case["blocks"]
    assert (b["src_off"][0::K] == b["src_off"][1::K]).all() and (b["pred_off"][0::K] != b["pred_off"][1::K]).all()
    rb = T.add_rate_info(case, 9, inter_share=0.7)
    o = T.oracle_tq_rd_batch(case, rb)
    g = T.hip_tq_rd_batch_device(ctx, case, null_recon=True)
    for n, x, y in list(zip(("recon", "qcoeff", "dqcoeff", "eob", "dist", "bits"), o, g))[1:]:
        assert np.array_equal(x, y), (n, int(np.sum(x != y)))
    # the candidates of a block really differ (different predictions -> different coefficients / costs)
    assert (o[5][0::K] != o[5][1::K]).mean() > 0.5 and (o[4][0::K, 0] != o[4][2::K, 0]).mean() > 0.5


def test_tq_rd_multi_reconstruction_buffers(ctx):
    """svt_hip_tq_rd_batch_multi_device: the blocks of one batch reconstruct into three separate buffers (pad_[0] bits 4-6 name the
    buffer) -- every output equals the oracle's, and each buffer holds exactly the reconstruction of its own blocks (elsewhere it
    keeps its fill)."""
    import torch
    lib = B.load()
    dev = torch.device("cuda", 0)
    case = T.make_tq_case(5, width=256, height=128)
    rb = T.add_rate_info(case, 3, 0.6)
    o = T.oracle_tq_rd_batch(case, rb)
    nb = len(case["blocks"])
    which = (np.arange(nb) * 7 // 3) % 3
    blocks = case["blocks"].copy()
    blocks["pad"][:, 0] |= (which << 4).astype(np.uint8)
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1)).to(dev)
    src, pred, dblk, qt, isc = up(case["src"]), up(case["pred"]), up(blocks), up(case["qtabs"]), up(case["iscan"])
    rtab, rscan = T.rate_tables()
    tab, scan = up(np.ascontiguousarray(rtab).reshape(1)), up(rscan)
    recs = [torch.full((case["src"].size,), 0x5A, dtype=torch.uint8, device=dev) for _ in range(3)]
    q, dq = torch.zeros(case["n_coeff"], dtype=torch.int16, device=dev), torch.zeros(case["n_coeff"], dtype=torch.int16, device=dev)
    eob, dist, bits = torch.zeros(nb, dtype=torch.int16, device=dev), torch.zeros(2 * nb, dtype=torch.int64, device=dev), torch.zeros(nb, dtype=torch.int32, device=dev)
    cnt = (C.c_int32 * 4)(*[int(v) for v in case["counts"]])
    p = lambda t: C.c_void_p(t.data_ptr())
    rset = (C.c_void_p * 3)(*[t.data_ptr() for t in recs])
    B.check(lib.svt_hip_tq_rd_batch_multi_device(ctx, p(src), p(pred), rset, 3, p(dblk), cnt, p(qt), p(isc), p(q), p(dq), p(eob), p(dist), p(tab), p(scan), p(bits)))
    B.check(lib.svt_hip_ctx_synchronize(ctx))
    assert np.array_equal(q.cpu().numpy(), o[1]) and np.array_equal(dq.cpu().numpy(), o[2]) and np.array_equal(eob.cpu().numpy().view(np.uint16), o[3])
    assert np.array_equal(dist.cpu().numpy().view(np.uint64).reshape(-1, 2), o[4]) and np.array_equal(bits.cpu().numpy(), o[5])
    W = case["src"].shape[1]
    own = np.full(case["src"].shape, -1, np.int32)
    for b, k in zip(case["blocks"], which):
        n = T.TX_N[int(b["tx_size"])]
        y, x = divmod(int(b["recon_off"]), W)
        own[y:y + n, x:x + n] = k
    for k in range(3):
        got = recs[k].cpu().numpy().reshape(case["src"].shape)
        assert np.array_equal(got[own == k], o[0][own == k]) and (got[own != k] == 0x5A).all(), k
    assert lib.svt_hip_tq_rd_batch_multi_device(ctx, p(src), p(pred), rset, 9, p(dblk), cnt, p(qt), p(isc), p(q), p(dq), p(eob), p(dist), p(tab), p(scan), p(bits)) == -1
    # only two of the three buffers passed: the blocks that name set 2 keep their coefficients but are reconstructed NOWHERE (not into
    # buffer 0 by default)
    for t in recs:
        t.fill_(0x5A)
    q.zero_()
    B.check(lib.svt_hip_tq_rd_batch_multi_device(ctx, p(src), p(pred), rset, 2, p(dblk), cnt, p(qt), p(isc), p(q), p(dq), p(eob), p(dist), p(tab), p(scan), p(bits)))
    B.check(lib.svt_hip_ctx_synchronize(ctx))
    assert np.array_equal(q.cpu().numpy(), o[1])
    for k in range(3):
        got = recs[k].cpu().numpy().reshape(case["src"].shape)
        if k < 2:
            assert np.array_equal(got[own == k], o[0][own == k]) and (got[own != k] == 0x5A).all(), k
        else:
            assert (got == 0x5A).all()
