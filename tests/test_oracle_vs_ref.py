"""CPU: the oracle (our restatement) against (a) the committed golden vectors produced by the reference
(tests/golden/*.npz, generator tests/gen_golden.py) and (b) the reference itself when oracle/_ref is present."""
import ctypes as C

import numpy as np
import pytest

import me_configs as MC
import svt_testlib as T

live = pytest.mark.skipif(not T.have_ref("libsvtref_kernels.so"), reason="oracle/_ref not built (reference sources absent)")


def _golden(name):
    return dict(np.load(f"{T.GOLDEN_DIR}/{name}"))


# ---------------- golden fixtures (always run, also on the GPU box) ----------------
def test_me_oracle_matches_reference_golden():
    g = _golden("me_reference.npz")
    assert len(g) == 9
    for key, ref in g.items():
        name, nl, tl, clip = key.split("|")
        nl, tl = int(nl), int(tl)
        gen = T.gen_clip if clip == "int" else T.gen_clip_subpel
        pics = [T.PaPic(f) for f in gen(264, 200, 3, 11)]
        res, _ = T.oracle_me_picture(pics[1], pics[0], pics[2] if nl == 2 else None, MC.preset(name, nl, tl))
        assert not T.me_results_equal(ref, res, nl), key


def test_tq_oracle_matches_reference_golden():
    g = _golden("tq_reference.npz")
    for seed in (1, 2):
        recon, q, dq, eob = T.oracle_tq_batch(T.make_tq_case(seed))
        assert np.array_equal(q, g[f"q{seed}"]) and np.array_equal(dq, g[f"dq{seed}"])
        assert np.array_equal(eob, g[f"eob{seed}"]) and np.array_equal(recon, g[f"recon{seed}"])


def test_lf_oracle_matches_reference_golden():
    g = _golden("lf_reference.npz")
    keys = sorted({k.split("|", 1)[1] for k in g})
    assert len(keys) == 3
    for k in keys:
        w, h, seed, sharp = (int(v) for v in k.split("|"))
        y, u, v = T.oracle_lf_frame(T.make_lf_case(seed, w, h, sharp))
        assert np.array_equal(y, g["y|" + k]) and np.array_equal(u, g["u|" + k]) and np.array_equal(v, g["v|" + k])


@pytest.mark.parametrize("seed", [1, 2])
def test_avg_ssd_oracle_matches_reference_golden(seed):
    """eb_vp9_combined_averaging_ssd (Codec/EbMotionEstimation.c:1708-1725), the quarter-pel metric of the SSD fractional
    search: the only leaf of that search with external linkage that does not go through the yasm-only Log2f"""
    g = _golden("avg_ssd_reference.npz")[str(seed)]
    o = T.oracle_avg_ssd_jobs(T.make_avg_ssd_jobs(seed))
    assert np.array_equal(o, g) and g.max() > 64 * 32 * 255 * 255 // 2


@pytest.mark.skipif(not T.have_ref("ref_me_sb"), reason="oracle/_ref/ref_me_sb not built (reference absent)")
def test_avg_ssd_oracle_matches_reference_live():
    jobs = T.make_avg_ssd_jobs(9, n=64)
    assert np.array_equal(T.oracle_avg_ssd_jobs(jobs), T.ref_avg_ssd_jobs(jobs))


def test_scan_tables_are_permutations():
    t = T.scan_tables()
    for ts in range(4):
        n = T.TX_N[ts] ** 2
        for tt in range(4):
            s, i = t[f"scan_{ts}_{tt}"], t[f"iscan_{ts}_{tt}"]
            assert sorted(s.tolist()) == list(range(n)) and np.array_equal(i[s], np.arange(n))


# ---------------- live comparison with the reference's own C kernels ----------------
@live
@pytest.mark.parametrize("ts", [0, 1, 2, 3])
def test_txfm_quant_inverse_vs_reference_kernels(ts):
    rng = np.random.default_rng(100 + ts)
    n = T.TX_N[ts]
    for tt in ([0, 1, 2, 3] if ts < 3 else [0]):
        for trial in range(24):
            kind = trial % 4
            if kind == 0:
                res = rng.integers(-255, 256, (n, n))
            elif kind == 1:
                res = rng.integers(-20, 21, (n, n))
            elif kind == 2:
                res = np.full((n, n), 255 if trial % 8 < 4 else -255)
            else:
                res = np.zeros((n, n), int)
                res[rng.integers(0, n), rng.integers(0, n)] = rng.integers(-255, 256)
            a, b = T.ref_fwd_txfm(res, ts, tt), T.oracle_fwd_txfm(res, ts, tt)
            assert np.array_equal(a, b), ("fwd", ts, tt, trial)
            if ts == 3:
                assert np.array_equal(T.ref_fwd_txfm(res, ts, tt, True), T.oracle_fwd_txfm(res, ts, tt, True))
            for qd, qa in ((4, 4), (40, 48), (200, 260), (1336, 1828)):
                qr = T.quant_table(qd, qa)
                rq, oq = T.ref_quantize(a, ts, tt, qr), T.oracle_quantize(a, ts, tt, qr)
                assert np.array_equal(rq[0], oq[0]) and np.array_equal(rq[1], oq[1]) and rq[2] == oq[2], ("quant", ts, tt, trial, qd)
                if rq[2]:
                    pred = rng.integers(0, 256, (n, n), dtype=np.uint8)
                    assert np.array_equal(T.ref_inv_add(rq[1], pred, ts, tt, rq[2]), T.oracle_inv_add(rq[1], pred, ts, tt, rq[2])), ("inv", ts, tt, trial, qd)


@live
def test_sad_leaf_kernels_vs_reference():
    import ctypes as C
    lib, ora = T.ref_kernels(), T.oracle()
    rng = np.random.default_rng(5)
    u8 = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint8))
    for trial in range(40):
        lowent = trial % 2 == 0  # low-entropy data forces ties: first minimum in raster order must win
        src = rng.integers(0, 4 if lowent else 256, (8, 16), dtype=np.uint8)
        sw, sh = int(rng.integers(1, 40)), int(rng.integers(1, 20))
        ref = rng.integers(0, 4 if lowent else 256, (sh + 16, sw + 24), dtype=np.uint8)
        out = []
        for fn in (lib.eb_vp9_sad_loop_kernel, ora.oracle_sad_loop):
            best, x, y = C.c_uint64(0), C.c_int16(-1), C.c_int16(-1)
            if fn is lib.eb_vp9_sad_loop_kernel:
                fn(u8(src), 16, u8(ref), 2 * ref.shape[1], 8, 16, C.byref(best), C.byref(x), C.byref(y), ref.shape[1], C.c_int16(sw), C.c_int16(sh))
            else:
                fn(u8(src), 16, u8(ref), 2 * ref.shape[1], 8, 16, C.byref(best), C.byref(x), C.byref(y), ref.shape[1], sw, sh)
            out.append((best.value, x.value, y.value))
        assert out[0] == out[1], (trial, out)
        a = rng.integers(0, 256, (16, 32), dtype=np.uint8)
        b = rng.integers(0, 256, (16, 32), dtype=np.uint8)
        c = rng.integers(0, 256, (16, 32), dtype=np.uint8)
        lib.eb_vp9_combined_averaging_sad.restype = C.c_uint32
        ora.oracle_avg_sad.restype = C.c_uint32
        assert lib.eb_vp9_combined_averaging_sad(u8(a), 32, u8(b), 32, u8(c), 32, 16, 32) == ora.oracle_avg_sad(u8(a), 32, u8(b), 32, u8(c), 32, 16, 32)


@live
@pytest.mark.parametrize("w,h,seed,sharp", [(192, 128, 1, 0), (200, 136, 2, 3), (136, 104, 6, 6), (264, 72, 9, 0)])
def test_lf_frame_vs_reference(w, h, seed, sharp):
    if not T.have_ref("ref_lf_frame"):
        pytest.skip("ref_lf_frame not built")
    case = T.make_lf_case(seed, w, h, sharp)
    for y_only in (False, True):
        o, r = T.oracle_lf_frame(case, y_only), T.ref_lf_frame(case, y_only)
        assert all(np.array_equal(a, b) for a, b in zip(o, r))


@live
def test_spatial_full_distortion_vs_reference_leaf():
    """the SSD metric of the fractional search (config C5): the oracle's leaf vs eb_vp9_spatial_full_distortion_kernel
    (the SSD search loop itself cannot run in the reference build: it indexes its table with the yasm-only Log2f)"""
    ref, ora = T.ref_kernels(), T.oracle()
    ref.eb_vp9_spatial_full_distortion_kernel.restype = C.c_uint64
    ora.oracle_spatial_full_distortion.restype = C.c_uint64
    rng = np.random.default_rng(33)
    for n in (8, 16, 32, 64):
        for lo, hi in ((0, 256), (0, 2), (254, 256)):
            a = rng.integers(lo, hi, (n, n + 8), dtype=np.uint8)
            b = rng.integers(0, 256, (n, n + 24), dtype=np.uint8) if lo == 0 else (255 - a[:, :n]).repeat(2, 1)[:, :n + 24].copy()
            want = ref.eb_vp9_spatial_full_distortion_kernel(a.ctypes.data_as(C.c_void_p), a.shape[1], b.ctypes.data_as(C.c_void_p), b.shape[1], n, n)
            got = ora.oracle_spatial_full_distortion(a.ctypes.data_as(C.c_void_p), a.shape[1], b.ctypes.data_as(C.c_void_p), b.shape[1], n, n)
            assert want == got, (n, lo, hi, want, got)


@live
def test_full_distortion_vs_reference_leaf():
    """T3: the oracle's coefficient-domain distortion vs the reference's full_distortion_kernel32bit, including
    differences that do not fit int16 (the reference squares them after an int16 truncation)."""
    ref = T.ref_kernels()
    rng = np.random.default_rng(21)
    for n in (4, 8, 16, 32):
        for hi in (300, 32767):
            a = rng.integers(-hi, hi + 1, (n, n)).astype(np.int16)
            b = rng.integers(-hi, hi + 1, (n, n)).astype(np.int16)
            want = np.zeros(2, np.uint64)
            ref.full_distortion_kernel32bit(a.ctypes.data_as(C.c_void_p), n, b.ctypes.data_as(C.c_void_p), n,
                                            want.ctypes.data_as(C.c_void_p), n, n)
            got = np.zeros(2, np.uint64)
            T.oracle().svt_oracle_full_distortion32(a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), n * n,
                                                    got.ctypes.data_as(C.c_void_p))
            assert np.array_equal(want, got), (n, hi, want, got)
