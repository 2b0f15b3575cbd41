"""The intra path of the oracle (oracle/oracle_intra.c) pinned against the reference's own code:
  * every predictor of VPX/intrapred.c, called directly (oracle/_ref/libsvtref_kernels.so), on random edges -- including the 4x4 forms
    that read the above-right samples;
  * whole intra pictures through oracle/_ref/ref_intra: generate_intra_reference_samples + intra_prediction + perform_coding_loop +
    the neighbour-array writer, block by block in the reference's coding order.
CPU only; skipped where the reference build is absent (the GPU box has the prebuilt files)."""
import ctypes as C

import numpy as np
import pytest

import svt_testlib as T
import encdec_model as M

NAMES = {1: "v", 2: "h", 3: "d45", 4: "d135", 5: "d117", 6: "d153", 7: "d207", 8: "d63", 9: "tm"}
DC = {(0, 0): "dc_128", (0, 1): "dc_top", (1, 0): "dc_left", (1, 1): "dc"}


def _oracle_pred(mode, bs, have_left, have_top, above, left):
    dst = np.zeros((bs, bs), np.uint8)
    T.oracle().svt_oracle_intra_predict(mode, bs, have_left, have_top, C.c_void_p(above.ctypes.data + 1), left.ctypes.data_as(C.c_void_p),
                                        dst.ctypes.data_as(C.c_void_p), bs)
    return dst


@pytest.mark.skipif(not T.have_ref("libsvtref_kernels.so"), reason="reference kernels not built")
@pytest.mark.parametrize("bs", [4, 8, 16, 32])
def test_predictors_match_reference(bs):
    ref = T.ref_kernels()
    rng = np.random.default_rng(100 + bs)
    for trial in range(40):
        above = rng.integers(0, 256, 2 * bs + 1 + 8, dtype=np.uint8)   # [0] = corner
        left = rng.integers(0, 256, bs, dtype=np.uint8)
        if trial % 5 == 0:
            above[:] = rng.choice([0, 255, 127]); left[:] = rng.choice([0, 255, 129])
        for mode, nm in NAMES.items():
            fn = getattr(ref, f"eb_vp9_{nm}_predictor_{bs}x{bs}_c")
            want = np.zeros((bs, bs), np.uint8)
            fn(want.ctypes.data_as(C.c_void_p), C.c_ssize_t(bs), C.c_void_p(above.ctypes.data + 1), left.ctypes.data_as(C.c_void_p))
            got = _oracle_pred(mode, bs, 1, 1, above, left)
            assert np.array_equal(got, want), (nm, bs, trial)
        for (hl, ht), nm in DC.items():
            fn = getattr(ref, f"eb_vp9_{nm}_predictor_{bs}x{bs}_c")
            want = np.zeros((bs, bs), np.uint8)
            fn(want.ctypes.data_as(C.c_void_p), C.c_ssize_t(bs), C.c_void_p(above.ctypes.data + 1), left.ctypes.data_as(C.c_void_p))
            assert np.array_equal(_oracle_pred(0, bs, hl, ht, above, left), want), (nm, bs, trial)


def _check_picture(W, H, seed, q_index, sizes=(8, 16, 32), modes=tuple(range(10))):
    src = T.gen_yuv(W, H, seed)
    mi = M.gen_intra_grid(seed, W, H, sizes=sizes, modes=modes)
    want = M.ref_intra_picture(src, mi, q_index)
    got = M.oracle_intra_picture(src, mi, q_index)
    for k in range(3):
        assert np.array_equal(got["pred"][k], want["pred"][k]), ("pred", k)
        assert np.array_equal(got["rec"].interior()[k], want["rec"][k]), ("rec", k)
    assert np.array_equal(got["qcoeff"], want["qcoeff"])
    assert np.array_equal(got["dqcoeff"], want["dqcoeff"])
    assert np.array_equal(got["eob_map"], want["eob_map"])
    return got


@pytest.mark.skipif(not T.have_ref("ref_intra"), reason="reference harness not built")
@pytest.mark.parametrize("W,H,seed,q", [(128, 64, 1, 60), (136, 72, 2, 120), (192, 128, 3, 20), (64, 200, 4, 200), (320, 192, 5, 255)])
def test_picture_matches_reference(W, H, seed, q):
    got = _check_picture(W, H, seed, q)
    assert got["eob_map"].any()


@pytest.mark.skipif(not T.have_ref("ref_intra"), reason="reference harness not built")
@pytest.mark.parametrize("W,H,seed,q,sizes", [(128, 64, 21, 60, (4,)), (136, 72, 22, 120, (4, 8)), (192, 128, 23, 30, (4, 8, 16, 32)), (72, 200, 24, 220, (4, 16))])
def test_pictures_with_4x4_blocks_match_reference(W, H, seed, q, sizes):
    """8x8 units of four 4x4 luma blocks: each with its own mode and transform type, the left ones reading true above-right samples"""
    got = _check_picture(W, H, seed, q, sizes=sizes)
    assert got["eob_map"].any()


@pytest.mark.skipif(not T.have_ref("ref_intra"), reason="reference harness not built")
@pytest.mark.parametrize("mode", range(10))
def test_4x4_single_mode(mode):
    _check_picture(96, 72, 60 + mode, 90, sizes=(4,), modes=(mode,))


@pytest.mark.skipif(not T.have_ref("ref_intra"), reason="reference harness not built")
@pytest.mark.parametrize("size", [8, 16, 32])
@pytest.mark.parametrize("mode", range(10))
def test_single_mode_pictures(size, mode):
    """one block size and one mode over a whole picture: every predictor at every picture-edge position (no left, no top, corner)"""
    _check_picture(128, 96, 40 + mode, 90, sizes=(8, size) if size > 8 else (8,), modes=(mode,))


def _square_inter_grid(seed, W, H):
    """an inter grid of square blocks 8x8 .. 64x64 (what the reference harness walks in mixed mode)"""
    rng = np.random.default_rng(seed)
    mi_rows, mi_cols = H // 8, W // 8
    mi = np.zeros((mi_rows, mi_cols), dtype=T.B.LF_MODE_INFO_DTYPE)

    def put(r, c, n8):
        b = mi[r:r + n8, c:c + n8]
        b["sb_type"], b["tx_size"], b["is_inter"], b["filter_level"] = {1: 3, 2: 6, 4: 9, 8: 12}[n8], min({1: 1, 2: 2, 4: 3, 8: 3}[n8], 3), 1, 20

    def split(r, c, n8):
        if r + n8 <= mi_rows and c + n8 <= mi_cols and (n8 == 1 or rng.random() < 0.4):
            return put(r, c, n8)
        h = n8 // 2
        for dr in (0, h):
            for dc in (0, h):
                if r + dr < mi_rows and c + dc < mi_cols:
                    split(r + dr, c + dc, h)
    for r in range(0, mi_rows, 8):
        for c in range(0, mi_cols, 8):
            split(r, c, 8)
    return mi


@pytest.mark.skipif(not T.have_ref("ref_intra"), reason="reference harness not built")
@pytest.mark.parametrize("W,H,seed,q", [(192, 128, 11, 100), (136, 72, 12, 40), (320, 192, 13, 200)])
def test_intra_blocks_of_inter_pictures_match_reference(W, H, seed, q):
    """mixed pictures: the inter blocks' reconstruction is given, the intra blocks predict from it exactly as the reference's
    neighbour-array bookkeeping makes them (blocks in its coding order, every block's reconstruction written when its turn comes)"""
    src = T.gen_yuv(W, H, seed)
    lf, _, n_intra = M.make_mixed(seed, _square_inter_grid(seed, W, H), share=0.35)
    assert n_intra > 5
    rng = np.random.default_rng(seed)
    inter_rec = [np.clip(p.astype(np.int32) + rng.integers(-6, 7, p.shape), 0, 255).astype(np.uint8) for p in src]
    want = M.ref_intra_picture_mixed(src, lf, q, inter_rec)
    rec = M.RefPic(W, H)
    for d, s_ in zip(rec.interior(), inter_rec):
        d[:] = s_
    got = M.oracle_intra_picture(src, lf, q, rec=rec, mixed=1)
    iy = np.kron(lf["is_inter"] == 0, np.ones((8, 8), bool))
    ic = np.kron(lf["is_inter"] == 0, np.ones((4, 4), bool))
    for k, m in enumerate((iy, ic, ic)):
        assert np.array_equal(got["pred"][k][m], want["pred"][k][m]), ("pred", k)
        assert np.array_equal(got["rec"].interior()[k], want["rec"][k]), ("rec", k)
        assert np.array_equal(got["rec"].interior()[k][~m], inter_rec[k][~m])                # inter blocks untouched
    assert np.array_equal(got["qcoeff"], want["qcoeff"]) and np.array_equal(got["dqcoeff"], want["dqcoeff"])
    assert np.array_equal(got["eob_map"], want["eob_map"]) and want["eob_map"].any()
