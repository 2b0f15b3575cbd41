"""GPU results against the REFERENCE itself, not against this repository's restatement of it: oracle/_ref holds the reference's own C
sources compiled in the build container (libsvtref_kernels.so: fwd_txfm.c, vp9_dct.c, quantize.c, inv_txfm.c, vp9_idct.c as they lie;
ref_me_sb: Codec/EbMotionEstimation.c + its C kernels) and travels to the GPU box prebuilt.  The transform stage is checked block by
block against eb_vp9_fht* / eb_vp9_fdct32x32 -> eb_vp9_quantize_b[_32x32] -> eb_vp9_idct*_add / eb_vp9_iht*_add on whole planes and
on saturating residuals (so the HIP butterflies are NOT only compared with the oracle's formulation of the same butterflies), motion
estimation against motion_estimate_sb on ~200 sampled superblocks of whole pictures with the three BASELINE presets."""
import ctypes as C
import os

import numpy as np
import pytest

import me_configs as MC
import svt_testlib as T
from test_gpu_me import hip_me_picture

B = T.B
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not T.have_ref("libsvtref_kernels.so"), reason="oracle/_ref not built (reference absent at build time)")]


@pytest.fixture(scope="module")
def ctx():
    lib = B.load()
    c = C.c_void_p()
    B.check(lib.svt_hip_ctx_create(C.byref(c), 0))
    yield c
    lib.svt_hip_ctx_destroy(c)


def reference_tq_blocks(case, recon_g, q_g, dq_g, eob_g):
    """every block of the case through the reference's own kernels; compares with the GPU outputs"""
    W = case["src"].shape[1]
    t = T.scan_tables()
    bad = []
    for i, k in enumerate(case["blocks"]):
        ts, tt, n = int(k["tx_size"]), int(k["tx_type"]), T.TX_N[int(k["tx_size"])]
        y, x = divmod(int(k["src_off"]), W)
        src, pred = case["src"][y:y + n, x:x + n], case["pred"][y:y + n, x:x + n]
        res = src.astype(np.int16) - pred.astype(np.int16)                        # eb_vp9_residual_kernel
        coeff = T.ref_fwd_txfm(res, ts, tt, bool(k["partial32"]))
        q, dq, eob = T.ref_quantize(coeff, ts, tt, case["qtabs"][int(k["qtab"])])
        co = int(k["coeff_off"])
        if not (np.array_equal(q, q_g[co:co + n * n]) and np.array_equal(dq, dq_g[co:co + n * n]) and eob == int(eob_g[i])):
            bad.append(("coeff", i, ts, tt))
            continue
        if k["do_recon"]:
            want = T.ref_inv_add(dq, pred, ts, tt, eob) if eob else pred          # eob 0: the prediction is the reconstruction
            if not np.array_equal(want, recon_g[y:y + n, x:x + n]):
                bad.append(("recon", i, ts, tt, eob))
    return bad


@pytest.mark.parametrize("seed,width,height,extreme,qsteps", [
    (21, 1920, 1088, False, ((40, 48), (8, 9), (200, 260))),         # a whole 1080p plane: ~30 k blocks of every size / type
    (22, 512, 256, True, ((4, 4), (1336, 1828), (40, 48))),          # +-255 residuals, smallest and largest quantiser steps
    (23, 512, 256, True, ((8, 9), (84, 100), (600, 800))),
])
def test_hip_transform_stage_vs_reference_kernels(ctx, seed, width, height, extreme, qsteps):
    case = T.make_tq_case(seed, width=width, height=height, extreme=extreme, qsteps=qsteps)
    recon, q, dq, eob = T.hip_tq_batch(ctx, case)
    bad = reference_tq_blocks(case, recon, q, dq, eob)
    assert not bad, (len(bad), bad[:6])
    assert len(set(eob.tolist())) > 8


def _sample_ranges(nsb, nsbx, want=200, run=20):
    """SB ranges spread over the picture: its first SBs, the end of a row / start of the next, the middle, the (incomplete) last row"""
    if nsb <= want:
        return [(0, nsb)]
    starts = [0, nsbx - run // 2, nsb // 3, nsb // 2 - run, nsb // 2 + nsbx // 2, 2 * nsb // 3, nsb - nsbx - run // 2, nsb - run]
    starts += [int(s) for s in np.linspace(nsbx * 3, nsb - nsbx * 3, max(0, want // run - len(starts)))]
    out = sorted({(max(0, s), min(nsb, max(0, s) + run)) for s in starts})
    return out


@pytest.mark.skipif(not T.have_ref("ref_me_sb"), reason="oracle/_ref/ref_me_sb not built")
@pytest.mark.parametrize("name,layer", [("c1_360p_m9", 1), ("c2_1080p_m8", 2), ("c3_2160p_m8", 3), ("c3_2160p_m8", 0), ("c5_sad", 1), ("c5_sad", 3)])
def test_hip_me_vs_reference_motion_estimate_sb(ctx, name, layer):
    """c5_sad: BASELINE C5's parameter set (2160p enc-mode 3 tune 0: four HME regions x three levels, 64x64 full-pel area, every PU
    refined) with the SAD fractional search instead of the SSD one -- the reference's SSD loop calls the yasm-only Log2f and cannot be
    built here, everything else C5 exercises runs in the reference's own motion_estimate_sb"""
    if name == "c5_sad":
        W, H = MC.PRESET_C5[1][:2]
        p = MC.preset_c5(2, layer)
        p.fractional_search_method = 0   # SVT_SUB_SAD_SEARCH
    else:
        W, H = MC.PRESETS[name][:2]
        p = MC.preset(name, 2, layer)
    frames = T.gen_clip_subpel(W, H, 3, 60 + layer)
    pics = [T.PaPic(f) for f in frames]
    g, _ = hip_me_picture(ctx, pics[1], pics[0], pics[2], p)
    nsb, nsbx = T.n_sb(W, H), (W + 63) // 64
    checked = 0
    for s0, s1 in _sample_ranges(nsb, nsbx):
        r, _ = T.ref_me_picture(pics[1], pics[0], pics[2], p, s0, s1)
        bad = T.me_results_equal(r[s0:s1], g[s0:s1], 2)
        assert not bad, (name, layer, s0, s1, bad)
        checked += s1 - s0
    assert checked >= min(nsb, 160)


@pytest.mark.skipif(not T.have_ref("ref_intra"), reason="oracle/_ref/ref_intra not built")
@pytest.mark.parametrize("W,H,seed,q,sizes", [(320, 192, 3, 100, (8, 16, 32)), (136, 200, 4, 40, (4, 8, 16, 32)), (704, 392, 5, 180, (4, 8, 16, 32))])
def test_hip_intra_pass_vs_reference_functions(ctx, W, H, seed, q, sizes):
    """the GPU's intra pass against the REFERENCE's own generate_intra_reference_samples + intra_prediction + perform_coding_loop +
    neighbour-array writer (oracle/_ref/ref_intra, travels to the box prebuilt), with no oracle in between: prediction, coefficients,
    eob of every block and the reconstruction before deblocking"""
    import encdec_model as M
    import test_gpu_intra as TI
    from test_gpu_encdec import flags_of
    lib = B.load()
    src = T.gen_yuv(W, H, seed)
    mi = M.gen_intra_grid(seed, W, H, sizes=sizes)
    flags = flags_of(enc_mode=8, tune=1, temporal_layer_index=0, is_used_as_reference=1, recon_file=0, loop_filter=0)   # no deblocking: what the harness returns
    assert not flags.apply_loop_filter
    thr = B.LfThresh()
    lib.svt_hip_lf_thresh_init(C.byref(thr), 0)
    rec = M.RefPic(W, H)
    g = TI.run_intra(ctx, src, mi, q, flags, thr, rec)
    assert g["rc"] == 0
    want = M.ref_intra_picture(src, mi, q)
    assert np.array_equal(g["pred"], np.concatenate([p.ravel() for p in want["pred"]])), "prediction"
    assert np.array_equal(g["q"], want["qcoeff"]) and np.array_equal(g["dq"], want["dqcoeff"]), "coefficients"
    assert np.array_equal(g["emap"], want["eob_map"]) and want["eob_map"].any(), "eob"
    for a, b in zip(rec.interior(g["rec"]), want["rec"]):
        assert np.array_equal(a, b), "reconstruction"
