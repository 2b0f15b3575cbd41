"""The reference's motion-estimation KERNEL PROCESS itself -- eb_vp9_motion_estimation_kernel (Codec/EbMotionEstimationProcess.c:875-1290) run as
a thread behind the reference's own FIFOs by oracle/_ref/ref_me_process -- against the oracle and the product's host forms (CPU), and against
the b-2 binding integration/me_process_binding.h executed where the reference calls (GPU, `-m gpu`).

What this pins that nothing else did: the wiring of the thread function around motion_estimate_sb (segments, SB buffers, the signal
derivation feeding the MeContext), and scope row M12's rate-control histograms (:1103-1237), which are inline in the thread function and
cannot be called in isolation."""
import ctypes as C
import os

import numpy as np
import pytest

import svt_testlib as T

B = T.B
needs_ref = pytest.mark.skipif(not T.have_ref("ref_me_process"), reason="oracle/_ref/ref_me_process not built (needs /root/reference)")

# (width, height, enc_mode, tune, temporal layer, P slice, used as reference, same_ref_poc, segments): BASELINE c1 / c2 / c3 parameter rows on
# pictures the reference's C path finishes in seconds; sizes are classes of eb_vp9_derive_input_resolution where the preset needs one
CASES = [
    (640, 360, 9, 1, 2, 0, 1, 0, (2, 2)),      # c1: 640x360 enc-mode 9 (whole picture)
    (640, 360, 9, 1, 0, 0, 1, 1, (1, 1)),      # c1 base layer: both lists on the same picture
    (320, 192, 8, 1, 4, 0, 0, 0, (3, 2)),      # enc-mode 8 parameters of the <= 576p class, deepest layer, not a reference
    (448, 256, 8, 1, 1, 1, 1, 0, (2, 1)),      # a P picture (one list)
]


def stats_for(nsb, seed):
    rng = np.random.default_rng(seed)
    var = rng.integers(0, 3000, (nsb, 5)).astype(np.uint16)
    var[::3, 1:] = rng.integers(0, 16257, (len(var[::3]), 4))
    var[:, 0] = rng.integers(0, 65536, nsb)
    cur_mean = rng.integers(0, 256, nsb).astype(np.uint8)
    ref_mean = np.clip(cur_mean.astype(np.int32) + rng.integers(-12, 13, nsb), 0, 255).astype(np.uint8)
    ref_var = np.clip(var[:, 0].astype(np.int64) + rng.integers(-60, 61, nsb), 0, 65535).astype(np.uint16)
    return dict(cur_mean=cur_mean, var=var, ref_mean=ref_mean, ref_var=ref_var)


def run_case(case, run_binding=False):
    W, H, mode, tune, tl, p_slice, used, same, seg = case
    fr = T.gen_clip_subpel(W, H, 3, 17 + W + tl)
    pics = [T.PaPic(f) for f in fr]
    nsb = T.n_sb(W, H)
    st = stats_for(nsb, 5 + W)
    ref1 = None if p_slice else (pics[0] if same else pics[2])
    out = T.ref_me_process(pics[1], pics[0], ref1, mode, tune, tl, p_slice=p_slice, used=used, rate_control_mode=1, same_ref_poc=same, segments=seg, stats=st,
                           run_binding=run_binding)
    return pics, ref1, st, out, nsb


@needs_ref
@pytest.mark.parametrize("case", CASES, ids=lambda c: f"{c[0]}x{c[1]}-m{c[2]}-tl{c[4]}-{'P' if c[5] else 'B'}")
def test_reference_process_vs_oracle_and_host_forms(case):
    W, H, mode, tune, tl, p_slice, used, same, seg = case
    pics, ref1, st, out, nsb = run_case(case)
    lib = B.load()
    # the parameters the binding reads out of the reference's control sets = what svt_hip_me_params_derive gives a host without them
    q = B.me_params_derive(pic_width=W, pic_height=H, enc_mode=mode, tune=tune, frame_rate=60, num_ref_lists=1 if p_slice else 2, temporal_layer_index=tl,
                           hierarchical_levels=4 if tune else 3, is_used_as_reference=used, same_ref_poc=same)
    q.rate_control_mode = 1
    assert bytes(out["binding_params"]) == bytes(q)
    # motion_estimate_sb through the thread function's SB loop, over all segments == the oracle
    ora, rcme = T.oracle_me_picture(pics[1], pics[0], ref1, out["binding_params"])
    assert not T.me_results_equal(ora, out["res"], 1 if p_slice else 2)
    assert np.array_equal(rcme, out["rcme"])
    # similar-collocated flags: host form of the product
    sim, sim_all = np.zeros(nsb, np.uint8), np.zeros(nsb, np.uint8)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    lib.svt_hip_me_similar_collocated(vp(st["cur_mean"]), vp(np.ascontiguousarray(st["var"][:, 0])), vp(st["ref_mean"]), vp(st["ref_var"]), nsb, 0, used, vp(sim), vp(sim_all))
    assert np.array_equal(sim, out["similar"]) and np.array_equal(sim_all, out["similar_all"])
    # stationary-edge part 1 + the rate-control SAD-interval indices and histograms (row M12; inline in the thread function): the oracle
    var85 = np.zeros((nsb, 85), np.uint16)
    var85[:, :5] = st["var"]
    p = B.MeSbStatsParams(W, H, lib.svt_hip_input_resolution(W, H), tl, 1 if p_slice else 0, 0, 1)
    o_stats, o_hist, o_full = T.oracle_me_sb_stats(dict(p=p, res=ora, var=var85, rcme=rcme, n=nsb))
    assert np.array_equal(o_stats["check1"], out["check1"]) and np.array_equal(o_stats["pm_check1"], out["pm_check1"])
    assert np.array_equal(o_stats["inter_idx"], out["inter_idx"]) and np.array_equal(o_stats["intra_idx"], out["intra_idx"])
    assert np.array_equal(o_hist[:128], out["me_hist"]) and np.array_equal(o_hist[128:], out["ois_hist"]) and o_full == out["full_sb_count"]


@needs_ref
@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES + [(1920, 1088, 8, 1, 3, 0, 1, 0, (4, 2)), (3840, 320, 8, 1, 2, 0, 1, 0, (6, 1))],
                         ids=lambda c: f"{c[0]}x{c[1]}-m{c[2]}-tl{c[4]}-{'P' if c[5] else 'B'}")
def test_binding_where_the_reference_calls(case):
    """integration/me_process_binding.h executed on the control sets the reference's own kernel process has just worked on: every MeCuResults
    record and rcme_distortion it writes equals what the reference's SB loop wrote (c1 / c2 / c3 parameter rows: the 1080p class at
    1920x1088, the 4K class on a 3840-wide strip)."""
    W, H, mode, tune, tl, p_slice, used, same, seg = case
    pics, ref1, st, out, nsb = run_case(case, run_binding=True)
    assert out["binding_rc"] == 0
    assert not T.me_results_equal(out["res"], out["binding_res"], 1 if p_slice else 2)
    assert np.array_equal(out["rcme"], out["binding_rcme"])
    # ... and the device form of the kernel process's bookkeeping against the reference's thread function
    import torch
    lib = B.load()
    ctx = C.c_void_p()
    B.check(lib.svt_hip_ctx_create(C.byref(ctx), 0))
    try:
        var85 = np.zeros((nsb, 85), np.uint16)
        var85[:, :5] = st["var"]
        p = B.MeSbStatsParams(W, H, lib.svt_hip_input_resolution(W, H), tl, 1 if p_slice else 0, 0, 1)
        g_stats, g_hist, g_full = T.hip_me_sb_stats(ctx, dict(p=p, res=out["binding_res"], var=var85, rcme=out["binding_rcme"], n=nsb))
    finally:
        lib.svt_hip_ctx_destroy(ctx)
    assert np.array_equal(g_stats["check1"], out["check1"]) and np.array_equal(g_stats["pm_check1"], out["pm_check1"])
    assert np.array_equal(g_stats["inter_idx"], out["inter_idx"]) and np.array_equal(g_stats["intra_idx"], out["intra_idx"])
    assert np.array_equal(g_hist[:128], out["me_hist"]) and np.array_equal(g_hist[128:], out["ois_hist"]) and g_full == out["full_sb_count"]
