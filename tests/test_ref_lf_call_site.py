"""The deblocking call site of the reference's encode pass (Codec/EbEncDecProcess.c:5676-5686: eb_vp9_build_mask_frame, then
eb_vp9_loop_filter_frame) run by the reference's own code on a real VP9_COMMON / MACROBLOCKD (oracle/_ref/ref_lf_binding) -- against the
oracle chain (CPU) and against the b-2 binding integration/loop_filter_binding.h executed in place of the second call (GPU, `-m gpu`)."""
import ctypes as C

import numpy as np
import pytest

import svt_testlib as T

B = T.B
needs_ref = pytest.mark.skipif(not T.have_ref("ref_lf_binding"), reason="oracle/_ref/ref_lf_binding not built (needs /root/reference)")
# (seed, width, height, filter level, sharpness, y_only)
CASES = [(1, 200, 136, 20, 0, False), (2, 640, 360, 33, 0, False), (3, 328, 200, 63, 3, False), (4, 256, 192, 9, 7, True), (5, 136, 72, 47, 0, False)]


def make(seed, W, H, level):
    case = T.make_lf_case(seed, W, H)              # blocky planes: the flat / flat2 paths are taken
    cells, _, mi = T.gen_mode_info_grid(seed, H // 8, W // 8, mi_stride=W // 8)
    mi = mi.copy()
    mi["filter_level"] = level                     # no deltas, no segmentation: every block filters at the frame's level (:286-289)
    return case, cells, mi


@needs_ref
@pytest.mark.parametrize("seed,W,H,level,sharp,y_only", CASES)
def test_reference_call_site_vs_oracle_chain(seed, W, H, level, sharp, y_only):
    case, cells, mi = make(seed, W, H, level)
    ref, _, _, lfm_ref = T.ref_lf_call_site(case["y"], case["u"], case["v"], cells, level, sharp, y_only)
    lfm = T.product_lf_build_masks(mi, H // 8, W // 8)
    for name in lfm.dtype.names:                   # eb_vp9_build_mask_frame == host form of the product (row L2)
        assert np.array_equal(lfm[name], lfm_ref[name]), name
    thr = B.LfThresh()
    B.load().svt_hip_lf_thresh_init(C.byref(thr), sharp)
    ora = T.oracle_lf_frame(dict(case, lfm=lfm, thr=thr), y_only)
    for name, a, b in zip("yuv", ref, ora):
        assert np.array_equal(a, b), name


@needs_ref
@pytest.mark.gpu
@pytest.mark.parametrize("seed,W,H,level,sharp,y_only", CASES + [(6, 1920, 1080, 26, 0, False), (7, 3840, 192, 40, 0, False)])
def test_binding_where_the_reference_calls(seed, W, H, level, sharp, y_only):
    case, cells, _ = make(seed, W, H, level)
    ref, bind, rc, _ = T.ref_lf_call_site(case["y"], case["u"], case["v"], cells, level, sharp, y_only, run_binding=True)
    assert rc == 0
    for name, a, b in zip("yuv", ref, bind):
        assert np.array_equal(a, b), name
