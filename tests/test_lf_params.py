"""Loop-filter parameter derivation (rows L0 / L1): svt_hip_lf_thresh_init against the reference's own eb_vp9_loop_filter_init
(sharpness 0..7), svt_hip_lf_level_from_q against eb_vp9_pick_filter_level (every base_qindex, inter and key frames).  The
committed fixture holds the reference's output; with oracle/_ref present the reference is run live.  The oracle's copies
(which generate the thresholds of every LF test case) are held to the same numbers."""
import ctypes as C
import os

import numpy as np
import pytest

import svt_testlib as T

B = T.B
GOLD = os.path.join(T.GOLDEN_DIR, "lf_params_reference.npz")


def _check(thr, pick):
    lib, ora = B.load(), T.oracle()
    assert thr.shape == (8, 3, 64) and pick.shape == (2, 256, 3)
    for sharp in range(8):
        for fn in (lib.svt_hip_lf_thresh_init, ora.svt_oracle_lf_thresh_init):
            t = B.LfThresh()
            fn(C.byref(t), sharp)
            got = np.array([list(t.mblim), list(t.lim), list(t.hev_thr)], np.uint8)
            assert np.array_equal(got, thr[sharp]), (sharp, fn)
    for key in (0, 1):
        for q in range(256):
            ac_q, level, sharp = (int(x) for x in pick[key, q])
            assert sharp == 0                       # eb_vp9_pick_filter_level always resets the sharpness
            assert lib.svt_hip_lf_level_from_q(ac_q, key) == level, (key, q)
            assert ora.svt_oracle_lf_level_from_q(ac_q, key) == level, (key, q)


def test_lf_params_vs_golden():
    g = np.load(GOLD)
    _check(g["thr"], g["pick"])
    # the AC step of every q index, as the quantiser fixture (eb_vp9_init_quantizer) has it
    quant = np.load(os.path.join(T.GOLDEN_DIR, "quant_reference.npz"))["0|0|0"]
    assert np.array_equal(quant[:, 3], g["pick"][0, :, 0])


@pytest.mark.skipif(not T.have_ref("ref_lf_frame"), reason="oracle/_ref/ref_lf_frame not built (reference absent)")
def test_lf_params_vs_reference_live():
    r = T.ref_lf_params()
    _check(r["thr"], r["pick"])
