"""GPU parity: HIP coefficient rate estimation (through the C ABI) vs the oracle and the golden fixture produced by the
reference's own coeff_rate_estimate(), bit-exact."""
import ctypes as C

import numpy as np
import pytest

import svt_testlib as T
from gen_golden import RATE_GOLDEN_CASES

B = T.B
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    lib = B.load()
    c = C.c_void_p()
    B.check(lib.svt_hip_ctx_create(C.byref(c), 0))
    yield c
    lib.svt_hip_ctx_destroy(c)


@pytest.mark.parametrize("seed,extreme", [(1, False), (2, False), (3, False), (4, False), (11, True), (12, True)])
def test_rate_vs_oracle(ctx, seed, extreme):
    case = T.make_rate_case(seed, extreme=extreme)
    o, g = T.oracle_rate_batch(case), T.hip_rate_batch(ctx, case)
    assert np.array_equal(o, g), (int((o != g).sum()), np.argwhere(o != g)[:6].ravel().tolist())


@pytest.mark.parametrize("seed,w,h,ext", RATE_GOLDEN_CASES)
def test_rate_vs_reference_golden(ctx, seed, w, h, ext):
    g = np.load(T.RATE_GOLD)
    case = T.make_rate_case(seed, width=w, height=h, extreme=ext)
    assert np.array_equal(T.hip_rate_batch(ctx, case), g[f"bits|{seed}|{w}|{h}|{int(ext)}"])


def test_rate_full_plane_and_order_independence(ctx):
    """every block of a 1920x1056 plane; shuffling the block list permutes the results"""
    case = T.make_rate_case(5, width=1920, height=1056)
    o, g = T.oracle_rate_batch(case), T.hip_rate_batch(ctx, case)
    assert np.array_equal(o, g)
    perm = np.random.default_rng(1).permutation(len(case["blocks"]))
    sh = dict(case)
    sh["blocks"] = case["blocks"][perm].copy()
    assert np.array_equal(T.hip_rate_batch(ctx, sh), g[perm])


def test_rate_rejects_bad_blocks(ctx):
    case = T.make_rate_case(1, width=64, height=64)
    case["blocks"]["ctx"][0] = 3
    with pytest.raises(RuntimeError):
        T.hip_rate_batch(ctx, case)
