"""(f)-2 inter prediction: the oracle (oracle/oracle_mc.c) against the reference's own inter_prediction() built from
/root/reference (oracle/_ref/ref_mc_frame, when present) and against the committed golden fixture it produced.
CPU only; the GPU parity test is tests/test_gpu_mc.py."""
import os

import numpy as np
import pytest

import svt_testlib as T
from gen_golden import MC_GOLDEN_CASES

GOLD = os.path.join(T.GOLDEN_DIR, "mc_reference.npz")
live = pytest.mark.skipif(not T.have_ref("ref_mc_frame"), reason="oracle/_ref/ref_mc_frame not built (reference absent)")


def _check(o, r, case):
    my, mc = T.mc_inter_masks(case)
    for name, a, b, m in zip("yuv", o, r, (my, mc, mc)):
        assert np.array_equal(a[m], b[m]), (name, int(np.sum((a != b) & m)))
        assert (a[~m] == 0x5A).all()  # units of non-inter blocks are left alone
    assert my.any()


@live
@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
@pytest.mark.parametrize("use_subpel", [1, 0])
def test_oracle_vs_reference(seed, use_subpel):
    case = T.make_mc_case(seed, use_subpel=use_subpel)
    _check(T.oracle_mc_frame(case), T.ref_mc_frame(case), case)


@live
def test_oracle_vs_reference_far_mvs_and_edges():
    """every MV far outside the picture: eb_vp9_clamp_mv_to_umv_border_sb decides all positions"""
    case = T.make_mc_case(11, width=136, height=72, mv_range=600)
    _check(T.oracle_mc_frame(case), T.ref_mc_frame(case), case)


@pytest.mark.parametrize("seed,w,h,sub", MC_GOLDEN_CASES)
def test_oracle_vs_golden(seed, w, h, sub):
    g = np.load(GOLD)
    case = T.make_mc_case(seed, width=w, height=h, use_subpel=sub)
    _check(T.oracle_mc_frame(case), [g[f"{p}|{seed}|{w}|{h}|{sub}"] for p in "yuv"], case)


def test_phase0_is_identity_and_compound_rounds_up():
    """zero MVs: single-list blocks copy the reference, compound blocks are (a + b + 1) >> 1"""
    case = T.make_mc_case(21, width=64, height=64, rect=False, intra_share=0.0)
    case["mi"]["mv_row"] = 0
    case["mi"]["mv_col"] = 0
    y, u, v = T.oracle_mc_frame(case)
    pad = case["pad"]
    r0, r1 = (case["refs"][l][0][pad:-pad, pad:-pad].astype(np.int32) for l in (0, 1))
    l0, l1 = np.kron(case["mi"]["ref_list"][:, :, 0], np.ones((8, 8), int)), np.kron(case["mi"]["ref_list"][:, :, 1], np.ones((8, 8), int))
    a = np.where(l0 == 0, r0, r1)
    b = np.where(l1 == 0, r0, r1)
    exp = np.where(l1 < 0, a, (a + b + 1) >> 1)
    assert np.array_equal(y, exp.astype(np.uint8))
