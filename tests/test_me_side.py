"""M12 per-SB side outputs: ZZ-SAD / non-moving index (GPU kernel vs oracle) and the similar-collocated flag (host C vs
oracle).  The oracle's ZZ-SAD is checked against an independent numpy formulation of compute_zz_sad and its SAD leaf
against the reference kernel when oracle/_ref is present."""
import ctypes as C

import numpy as np
import pytest

import svt_testlib as T

B = T.B


def _numpy_zz(cur, prev, shift):
    h, w = prev.shape
    ny, nx = (h + 63) // 64, (w + 63) // 64
    zz, nmi = np.zeros(ny * nx, np.uint32), np.zeros(ny * nx, np.uint8)
    c16 = cur[::4, ::4].astype(np.int32)
    for sy in range(ny):
        for sx in range(nx):
            sb = sy * nx + sx
            if sx * 64 + 64 <= w and sy * 64 + 64 <= h:
                d = prev[sy * 64:sy * 64 + 64:4, sx * 64:sx * 64 + 64:4].astype(np.int32)
                v = int(np.abs(c16[sy * 16:sy * 16 + 16, sx * 16:sx * 16 + 16] - d).sum())
                base = 256
            else:
                v = 0xFFFFFFFF
                base = (min(64, w - sx * 64) >> 2) * (min(64, h - sy * 64) >> 2)
            zz[sb] = v
            nmi[sb] = 0 if v < (base * 2) >> shift else 10 if v < (base * 4) >> shift else 20 if v < (base * 8) >> shift else 30
    return zz, nmi


@pytest.mark.parametrize("res", [0, 2, 3])
def test_oracle_zz_sad_vs_numpy(res):
    f = T.gen_clip(328, 200, 2, 31)
    f[1][:64, :128] = f[0][:64, :128]            # two static SBs -> score 0
    f[1][64:128, :64] = np.clip(f[0][64:128, :64].astype(np.int16) + 3, 0, 255).astype(np.uint8)
    zz, nmi = T.oracle_me_zz_sad(T.PaPic(f[1]), T.PaPic(f[0]), res)
    wz, wn = _numpy_zz(f[1], f[0], [4, 2, 0, 0][res])
    assert np.array_equal(zz, wz) and np.array_equal(nmi, wn)
    assert zz[0] == 0 and nmi[0] == 0 and (zz == 0xFFFFFFFF).any() and len(set(nmi.tolist())) >= 2


def test_similar_collocated_host_vs_oracle():
    rng = np.random.default_rng(4)
    n = 4000
    cm, rm = rng.integers(0, 256, n).astype(np.uint8), rng.integers(0, 256, n).astype(np.uint8)
    rm[: n // 2] = np.clip(cm[: n // 2].astype(np.int16) + rng.integers(-12, 13, n // 2), 0, 255).astype(np.uint8)
    cv = rng.integers(0, 3000, n).astype(np.uint16)
    rv = np.clip(cv.astype(np.int32) + rng.integers(-40, 41, n), 0, 65535).astype(np.uint16)
    rv[::7] = 0
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    for i_slice in (0, 1):
        for is_ref in (0, 1):
            a, b, c, d = (np.zeros(n, np.uint8) for _ in range(4))
            B.load().svt_hip_me_similar_collocated(vp(cm), vp(cv), vp(rm), vp(rv), n, i_slice, is_ref, vp(a), vp(b))
            T.oracle().svt_oracle_me_similar_collocated(vp(cm), vp(cv), vp(rm), vp(rv), n, i_slice, is_ref, vp(c), vp(d))
            assert np.array_equal(a, c) and np.array_equal(b, d)
            if not i_slice:
                assert 0 < b.sum() < n and (a.sum() > 0) == bool(is_ref)


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,res", [(328, 200, 0), (1920, 1080, 2), (3840, 2160, 3)])
def test_gpu_zz_sad_vs_oracle(w, h, res):
    lib = B.load()
    ctx = C.c_void_p()
    B.check(lib.svt_hip_ctx_create(C.byref(ctx), 0))
    try:
        f = T.gen_clip(w, h, 2, 33)
        f[1][:64, :192] = f[0][:64, :192]
        cur, prev = T.PaPic(f[1]), T.PaPic(f[0])
        o = T.oracle_me_zz_sad(cur, prev, res)
        g = T.hip_me_zz_sad(ctx, cur, prev, res)
        assert np.array_equal(o[0], g[0]) and np.array_equal(o[1], g[1])
    finally:
        lib.svt_hip_ctx_destroy(ctx)


@pytest.mark.skipif(not T.have_ref("ref_me_side"), reason="oracle/_ref/ref_me_side not built (reference absent)")
@pytest.mark.parametrize("res,w,h", [(0, 328, 200), (2, 384, 256), (3, 640, 384), (1, 200, 72)])
def test_oracle_and_host_vs_reference_me_side(res, w, h):
    """compute_zz_sad and eb_vp9_derive_similar_collocated_flag of the reference built from source: non-moving index per SB
    (incl. incomplete SBs) and both similarity flags, for I / non-I slices and referenced / unreferenced pictures"""
    f = T.gen_clip(w, h, 2, 31 + res)
    f[1][:64, :128] = f[0][:64, :128]
    f[1][64:128, :64] = np.clip(f[0][64:128, :64].astype(np.int16) + 3, 0, 255).astype(np.uint8)
    cur, prev = T.PaPic(f[1]), T.PaPic(f[0])
    n = T.n_sb(w, h)
    rng = np.random.default_rng(5 + res)
    cm = rng.integers(0, 256, n).astype(np.uint8)
    rm = np.clip(cm.astype(np.int16) + rng.integers(-14, 15, n), 0, 255).astype(np.uint8)
    cv = rng.integers(0, 3000, n).astype(np.uint16)
    rv = np.clip(cv.astype(np.int32) + rng.integers(-40, 41, n), 0, 65535).astype(np.uint16)
    rv[::5] = 0
    _, nmi = T.oracle_me_zz_sad(cur, prev, res)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    for i_slice in (0, 1):
        for is_ref in (0, 1):
            r_nmi, r_sim, r_all = T.ref_me_side(cur, prev, res, cm, cv, rm, rv, i_slice, is_ref)
            assert np.array_equal(r_nmi, nmi)
            a, b = np.zeros(n, np.uint8), np.zeros(n, np.uint8)
            B.load().svt_hip_me_similar_collocated(vp(cm), vp(cv), vp(rm), vp(rv), n, i_slice, is_ref, vp(a), vp(b))
            assert np.array_equal(a, r_sim) and np.array_equal(b, r_all)
            T.oracle().svt_oracle_me_similar_collocated(vp(cm), vp(cv), vp(rm), vp(rv), n, i_slice, is_ref, vp(a), vp(b))
            assert np.array_equal(a, r_sim) and np.array_equal(b, r_all)
    assert len(set(nmi.tolist())) >= 2


# ---- M12, rest: stationary-edge flags and rate-control SAD-interval indices / histograms ----
def _flags(out):
    return np.stack([out["check1"], out["pm_check1"], out["check2"], out["low_dist_logo"]], axis=1)


@pytest.mark.parametrize("c", T.SB_STATS_CASES)
def test_sb_stats_oracle_vs_reference_golden(c):
    case = T.make_sb_stats_case(*c)
    out, hist, full = T.oracle_me_sb_stats(case)
    g = np.load(f"{T.GOLDEN_DIR}/sb_stats_reference.npz")[str(c[0])]
    assert np.array_equal(_flags(out), g[:, :4]), c
    nx = (c[1] + 63) // 64
    complete = np.array([(sb % nx) * 64 + 64 <= c[1] and (sb // nx) * 64 + 64 <= c[2] for sb in range(case["n"])])
    assert np.array_equal(complete, g[:, 5].astype(bool))
    assert g[:, 4].any() and not g[:, 4].all() or case["n"] < 40           # some SBs can hold a logo, not all
    # histograms: one count per complete SB, the indices they were counted under
    assert full == complete.sum() and hist[128:].sum() == full
    assert hist[:128].sum() == (full if c[5] != 2 else 0)
    assert np.array_equal(np.bincount(out["intra_idx"][complete], minlength=128), hist[128:])
    assert out["inter_idx"].max() <= 127 and (out["inter_idx"][~complete] == 0).all()


@pytest.mark.skipif(not T.have_ref("ref_me_side"), reason="oracle/_ref/ref_me_side not built (reference absent)")
@pytest.mark.parametrize("c", [(11, 3840, 2160, 3, 1, 0, 1), (12, 1280, 720, 1, 0, 2, 1), (13, 720, 576, 0, 2, 0, 0)])
def test_sb_stats_oracle_vs_reference_live(c):
    case = T.make_sb_stats_case(*c)
    out, _, _ = T.oracle_me_sb_stats(case)
    r = T.ref_me_stationary_edge(case)
    assert np.array_equal(_flags(out), r[:, :4])
    assert len(set(map(tuple, r[:, :4].tolist()))) >= 3


@pytest.fixture(scope="module")
def gctx():
    lib = B.load()
    ctx = C.c_void_p()
    B.check(lib.svt_hip_ctx_create(C.byref(ctx), 0))
    yield ctx
    lib.svt_hip_ctx_destroy(ctx)


@pytest.mark.gpu
@pytest.mark.parametrize("c", T.SB_STATS_CASES)
def test_gpu_sb_stats_flags_vs_oracle_and_reference_golden(gctx, c):
    """PINNED part of M12: the stationary-edge / logo flags (the reference's static part1 / part2 functions + its
    eb_vp9_sb_params_init, golden fixture sb_stats_reference.npz)."""
    case = T.make_sb_stats_case(*c)
    o, _, _ = T.oracle_me_sb_stats(case)
    g, _, _ = T.hip_me_sb_stats(gctx, case)
    assert np.array_equal(_flags(o), _flags(g))
    assert np.array_equal(_flags(g), np.load(f"{T.GOLDEN_DIR}/sb_stats_reference.npz")[str(c[0])][:, :4])


@pytest.mark.gpu
@pytest.mark.parametrize("c", T.SB_STATS_CASES)
def test_gpu_sb_stats_histograms_vs_oracle_parity_unpinned(gctx, c):
    """UNPINNED part of M12: the rate-control SAD-interval indices and histograms.  That arithmetic is inline in the
    reference's thread function (Codec/EbMotionEstimationProcess.c:1103-1237) and cannot be called in isolation, so the oracle
    restates it by reading: HIP == oracle here proves that two restatements agree, not parity with the reference."""
    case = T.make_sb_stats_case(*c)
    o, oh, of = T.oracle_me_sb_stats(case)
    g, gh, gf = T.hip_me_sb_stats(gctx, case)
    assert np.array_equal(o["inter_idx"], g["inter_idx"]) and np.array_equal(o["intra_idx"], g["intra_idx"])
    assert np.array_equal(oh, gh) and of == gf
