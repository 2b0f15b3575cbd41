"""Deblocked reconstruction -> reference-picture form (scope: the hand-over between the loop filter and the inter prediction of
the next temporal layer): pad_ref_and_set_flags (Codec/EbEncDecProcess.c:4822-4851) -> eb_vp9_generate_padding
(Codec/EbMcp.c:17-58) on Y / Cb / Cr, in place.

oracle vs numpy's edge padding, vs the reference's own function run here (oracle/_ref/ref_refpad, when present) and vs the
committed outputs of that run (tests/golden/refpad_reference.npz); svt_hip_ref_pad_batch_device vs the oracle and the golden
outputs on the GPU."""
import ctypes as C
import os

import numpy as np
import pytest

import svt_testlib as T

B = T.B
GOLD = os.path.join(T.GOLDEN_DIR, "refpad_reference.npz")


def _numpy_pad(case):
    out = []
    for b, sh in zip(case["bufs"], (0, 1, 1)):
        px, py, w, h = case["pad_x"] >> sh, case["pad_y"] >> sh, case["width"] >> sh, case["height"] >> sh
        out.append(np.pad(b[py:py + h, px:px + w], ((py, py), (px, px)), mode="edge"))
    return out


@pytest.mark.parametrize("args", T.REFPAD_GOLDEN_CASES + ((9, 640, 360, 80, 80, 0),))
def test_oracle_vs_numpy(args):
    case = T.make_refpad_case(*args)
    for a, b in zip(T.refpad_valid(case, T.oracle_ref_pad(case)), _numpy_pad(case)):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("args", T.REFPAD_GOLDEN_CASES)
def test_oracle_vs_golden(args):
    gold = np.load(GOLD)
    case = T.make_refpad_case(*args)
    for k, (a, g) in enumerate(zip(T.oracle_ref_pad(case), (gold[f"{args[0]}|{k}"] for k in range(3)))):
        assert np.array_equal(T.refpad_valid(case, [a])[0], T.refpad_valid(case, [g])[0]), k


@pytest.mark.skipif(not T.have_ref("ref_refpad"), reason="oracle/_ref/ref_refpad not built")
@pytest.mark.parametrize("args", T.REFPAD_GOLDEN_CASES + ((7, 328, 200, 80, 80, 0),))
def test_oracle_vs_reference(args):
    case = T.make_refpad_case(*args)
    for a, b in zip(T.oracle_ref_pad(case), T.ref_ref_pad(case)):
        assert np.array_equal(a, b)   # whole buffers: the reference copies `stride` bytes per border row, the oracle too


@pytest.fixture(scope="module")
def ctx():
    lib = B.load()
    c = C.c_void_p()
    B.check(lib.svt_hip_ctx_create(C.byref(c), 0))
    yield c
    lib.svt_hip_ctx_destroy(c)


@pytest.mark.gpu
@pytest.mark.parametrize("args", T.REFPAD_GOLDEN_CASES + ((11, 3840, 2160, 80, 80, 0), (12, 1920, 1080, 80, 80, 0), (13, 640, 360, 80, 80, 0),
                                                          (14, 8, 8, 80, 80, 0), (15, 66, 34, 6, 2, 1)))
def test_gpu_vs_oracle(ctx, args):
    case = T.make_refpad_case(*args)
    got = T.hip_ref_pad_batch(ctx, [case])[0]
    want = T.oracle_ref_pad(case)
    for k, (a, b, orig) in enumerate(zip(got, want, case["bufs"])):
        assert np.array_equal(T.refpad_valid(case, [a])[0], T.refpad_valid(case, [b])[0]), k
        if case["slack"]:   # the kernel writes nothing behind the padded row
            assert np.array_equal(a[:, -case["slack"]:], orig[:, -case["slack"]:]), k


@pytest.mark.gpu
def test_gpu_vs_golden(ctx):
    gold = np.load(GOLD)
    for args in T.REFPAD_GOLDEN_CASES:
        case = T.make_refpad_case(*args)
        got = T.hip_ref_pad_batch(ctx, [case])[0]
        for k in range(3):
            assert np.array_equal(T.refpad_valid(case, [got[k]])[0], T.refpad_valid(case, [gold[f"{args[0]}|{k}"]])[0]), (args, k)


@pytest.mark.gpu
def test_gpu_batch_of_different_sizes(ctx):
    cases = [T.make_refpad_case(20 + i, w, h, 80, 80, 0) for i, (w, h) in enumerate(((640, 360), (72, 40), (1000, 568), (64, 64)))]
    for case, got in zip(cases, T.hip_ref_pad_batch(ctx, cases)):
        for a, b in zip(got, T.oracle_ref_pad(case)):
            assert np.array_equal(a, b)


@pytest.mark.gpu
def test_gpu_rejects_bad_geometry(ctx):
    lib = B.load()
    case = T.make_refpad_case(1, 64, 64, 16, 16)
    d = (B.YuvPlanes * 1)(T._refpad_desc(case, [b.ctypes.data for b in case["bufs"]]))
    d[0].y_stride = 64 + 16   # smaller than the padded row
    assert lib.svt_hip_ref_pad_batch_device(ctx, 1, d, 16, 16) == -1
    assert lib.svt_hip_ref_pad_batch_device(ctx, 1, d, 15, 16) == -1
