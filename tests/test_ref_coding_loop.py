"""The transform call sites of the reference's encode pass (Codec/EbEncDecProcess.c:3830, 3890, 3940 -> perform_coding_loop :365-587) run by
the reference's own code with its EncDecContext / MACROBLOCKD / ModeInfo / QUANTS around them (oracle/_ref/ref_tq_binding) -- against the
oracle (CPU) and against the b-2 binding integration/coding_loop_binding.h executed at the same call sites (GPU, `-m gpu`)."""
import ctypes as C

import numpy as np
import pytest

import svt_testlib as T

B = T.B
needs_ref = pytest.mark.skipif(not T.have_ref("ref_tq_binding"), reason="oracle/_ref/ref_tq_binding not built (needs /root/reference)")
CASES = [(1, 256, 192, 160), (2, 320, 192, 60), (3, 192, 128, 220), (4, 448, 256, 120)]   # seed, width, height, q index


def make(seed, W, H):
    rng = np.random.default_rng(seed)
    y = T.gen_clip(W, H, 1, seed)[0]
    src = (y, (y[::2, ::2] // 2 + 32).astype(np.uint8), (255 - y[::2, ::2] // 2 - y[1::2, 1::2] // 4).astype(np.uint8))
    pred = []
    for p_ in src:
        q = np.clip(np.roll(p_, (1, 2), (0, 1)).astype(np.int16) + rng.integers(-10, 11, p_.shape), 0, 255).astype(np.uint8)
        q[: p_.shape[0] // 4] = p_[: p_.shape[0] // 4]                       # a band with zero residual: eob 0, the reduced inverse paths
        pred.append(q)
    # square inter blocks: every 64x64 area is one of {64x64 (four 32x32 units), 32x32, 16x16, 8x8, 4x4}
    mi = np.zeros((H // 8, W // 8), dtype=B.LF_MODE_INFO_DTYPE)
    mi["is_inter"] = 1
    for r in range(0, H // 8, 8):
        for c in range(0, W // 8, 8):
            k = int(rng.integers(0, 5))
            if r + 8 > H // 8 or c + 8 > W // 8:
                k = min(k, 3) if (H // 8 - r) % 4 == 0 and (W // 8 - c) % 4 == 0 else min(k, 2)
            sb_type, tx = [(0, 0), (3, 1), (6, 2), (9, 3), (12, 3)][k]
            mi["sb_type"][r:r + 8, c:c + 8] = sb_type
            mi["tx_size"][r:r + 8, c:c + 8] = tx
    return src, tuple(pred), mi


def oracle_side(src, pred, ref, q_index):
    """the oracle's transform batch over the blocks the reference visited, in the reference's order"""
    lib = B.load()
    H, W = src[0].shape
    ny, nc = W * H, W * H // 4
    po = (0, ny, ny + nc)
    srcp, predp = (np.concatenate([p_.ravel() for p_ in s]) for s in (src, pred))
    iscan, offs = T.iscan_array()
    qt = np.zeros(2, dtype=B.QUANT_DTYPE)
    tabs = (B.QuantTables * 2)() if hasattr(B, "QuantTables") else None
    assert lib.svt_hip_quant_tables_for_qindex(q_index, qt.ctypes.data_as(C.c_void_p)) == 0
    blk = ref["blocks"]
    arr = np.zeros(len(blk), dtype=B.TQ_BLOCK_DTYPE)
    pos = 0
    for i, b in enumerate(blk):
        ps = W // 2 if b["plane"] else W
        o = po[b["plane"]] + int(b["y"]) * ps + int(b["x"])
        arr[i] = (o, o, o, pos, offs[(int(b["tx_size"]), 0)], ps, ps, ps, b["tx_size"], 0, 1 if b["plane"] else 0, 1, 0, 0)
        pos += 16 << (2 * int(b["tx_size"]))
    case = dict(src=srcp, pred=predp, blocks=arr, qtabs=qt, iscan=iscan, n_coeff=pos)
    recon, q, dq, eob = T.oracle_tq_batch(case)
    return recon, q, dq, eob


@needs_ref
@pytest.mark.parametrize("seed,W,H,q_index", CASES)
def test_reference_call_sites_vs_oracle(seed, W, H, q_index):
    src, pred, mi = make(seed, W, H)
    ref, _, _ = T.ref_coding_loop_call_sites(src, pred, mi, q_index)
    recon, q, dq, eob = oracle_side(src, pred, ref, q_index)
    assert set(np.unique(ref["blocks"]["tx_size"])) >= {0, 1, 2} and (ref["blocks"]["eob"] == 0).any() and (ref["blocks"]["eob"] > 0).any()
    assert np.array_equal(eob, ref["blocks"]["eob"])
    assert np.array_equal(q, ref["q"]) and np.array_equal(dq, ref["dq"])
    ny, nc = W * H, W * H // 4
    for name, a, b in zip("yuv", ref["rec"], (recon[:ny], recon[ny:ny + nc], recon[ny + nc:])):
        assert np.array_equal(a.ravel(), b), name


@needs_ref
@pytest.mark.gpu
@pytest.mark.parametrize("seed,W,H,q_index", CASES + [(5, 1920, 1088, 160)])
def test_binding_where_the_reference_calls(seed, W, H, q_index):
    src, pred, mi = make(seed, W, H)
    ref, bind, rc = T.ref_coding_loop_call_sites(src, pred, mi, q_index, run_binding=True)
    assert rc == 0
    assert np.array_equal(ref["blocks"], bind["blocks"])          # same blocks, same eobs
    assert np.array_equal(ref["q"], bind["q"]) and np.array_equal(ref["dq"], bind["dq"])
    for name, a, b in zip("yuv", ref["rec"], bind["rec"]):
        assert np.array_equal(a, b), name
