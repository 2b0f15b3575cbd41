"""(f)-4 coefficient rate estimation: the oracle (oracle/oracle_rate.c) against the reference's own coeff_rate_estimate()
built from source (oracle/_ref/ref_rate_blocks, when present) and against the committed golden fixture (which also holds
the reference's cost tables and scan orders -- the inputs of the GPU path).  CPU only; GPU parity: tests/test_gpu_rate.py."""
import numpy as np
import pytest

import svt_testlib as T
from gen_golden import RATE_GOLDEN_CASES

live = pytest.mark.skipif(not T.have_ref("ref_rate_blocks"), reason="oracle/_ref/ref_rate_blocks not built (reference absent)")


@live
@pytest.mark.parametrize("seed,extreme", [(1, False), (2, False), (3, False), (4, False), (11, True), (12, True)])
def test_oracle_vs_reference(seed, extreme):
    case = T.make_rate_case(seed, extreme=extreme)
    ref_bits, tab, scan = T.ref_rate_run(case)
    assert np.array_equal(T.oracle_rate_batch(case, tab, scan), ref_bits)
    if not extreme:  # the ordinary cases really spread over the eob classes
        assert len(set(case["blocks"]["eob"].tolist())) > 8 and (case["blocks"]["eob"] == 0).any()


@live
def test_golden_tables_are_the_reference_tables():
    dummy = dict(qcoeff=np.zeros(16, np.int16), blocks=np.zeros(1, dtype=T.B.RATE_BLOCK_DTYPE), tx_type=np.zeros(1, np.int32))
    _, tab, scan = T.ref_rate_run(dummy)
    gt, gs = T.rate_tables()
    assert np.array_equal(scan, gs) and all(np.array_equal(tab[k], gt[k]) for k in ("token_costs", "value_cost", "cat6_low_cost", "cat6_high_cost"))


@pytest.mark.parametrize("seed,w,h,ext", RATE_GOLDEN_CASES)
def test_oracle_vs_golden(seed, w, h, ext):
    g = np.load(T.RATE_GOLD)
    case = T.make_rate_case(seed, width=w, height=h, extreme=ext)
    assert np.array_equal(T.oracle_rate_batch(case), g[f"bits|{seed}|{w}|{h}|{int(ext)}"])


def test_full_block_has_no_eob_token_and_empty_block_is_one_token():
    """eob == n: the EOB token is not coded; eob == 0: the cost is the EOB token of band 0 in the given context"""
    tab, scan = T.rate_tables()
    offs, _ = T.rate_scan_offsets()
    blocks = np.zeros(6, dtype=T.B.RATE_BLOCK_DTYPE)
    q = np.ones(16 * 3 + 64 * 3, np.int16)
    for i in range(3):
        blocks[i] = (16 * i, offs[(0, 0)], (16, 15, 0)[i], 0, 0, 0, i, (0, 0))
        blocks[3 + i] = (48 + 64 * i, offs[(1, 0)], (64, 63, 0)[i], 1, 1, 1, i, (0, 0))
    q[16 + 15] = 0
    q[48 + 64 + int(scan[offs[(1, 0)] + 63])] = 0
    bits = T.oracle_rate_batch(dict(qcoeff=q, blocks=blocks))
    assert bits[2] == tab["token_costs"][0, 0, 0, 0, 0, 2, 11] and bits[5] == tab["token_costs"][1, 1, 1, 0, 0, 2, 11]
    assert bits[0] != bits[1] and bits[3] != bits[4]


def test_compiled_in_4x4_scan_orders_are_the_reference_tables():
    """the fused distortion + rate entry walks 4x4 blocks with scan orders compiled into the kernel: they must be the
    reference's eb_vp9_scan_orders[TX_4X4][tx_type] (scan + neighbours), as the committed fixture has them"""
    import ctypes as C
    _, scan = T.rate_tables()
    offs, _ = T.rate_scan_offsets()
    for tt in range(4):
        out = np.zeros(48, np.int16)
        assert T.B.load().svt_hip_rate_scan4x4_table(tt, out.ctypes.data_as(C.c_void_p)) == 0
        o = offs[(0, tt)]
        assert np.array_equal(out[:16], scan[o:o + 16]) and np.array_equal(out[16:], scan[o + 16:o + 48]), tt
    assert T.B.load().svt_hip_rate_scan4x4_table(4, None) != 0
