"""Picture-analysis pre-ME stage ("next" row f-1): padded full / quarter / sixteenth planes from the input luma.

oracle vs the reference's eb_vp9_decimation_2d + eb_vp9_generate_padding (oracle/_ref/libsvtref_pa.so, when present), vs an
independent numpy construction (which is what the ME tests feed), and the HIP kernel vs the oracle on the GPU."""
import ctypes as C
import os

import numpy as np
import pytest

import svt_testlib as T

B = T.B
PADS = (68, 32, 16)


def _alloc(w, h):
    planes = [np.full((h // s + 2 * p, w // s + 2 * p), 0xCD, np.uint8) for s, p in zip((1, 2, 4), PADS)]
    d = B.PaPicture()
    for name, a, p in zip(("full", "quarter", "sixteenth"), planes, PADS):
        setattr(d, name, B.plane_desc(a, p, p))
    return planes, d


def _numpy_planes(luma):
    return [np.pad(luma[::s, ::s], p, mode="edge") for s, p in zip((1, 2, 4), PADS)]


@pytest.mark.parametrize("w,h", [(64, 64), (328, 200), (640, 360)])
def test_oracle_vs_numpy(w, h):
    luma = T.gen_clip(w, h, 1, 3)[0]
    planes, d = _alloc(w, h)
    assert T.oracle().svt_oracle_pa_prepare(luma.ctypes.data_as(C.c_void_p), luma.strides[0], C.byref(d), 1) == 0
    for a, b in zip(planes, _numpy_planes(luma)):
        assert np.array_equal(a, b)


@pytest.mark.skipif(not os.path.exists(os.path.join(T.REF_DIR, "libsvtref_pa.so")), reason="oracle/_ref/libsvtref_pa.so not built")
def test_oracle_vs_reference_leaf_functions():
    ref = C.CDLL(os.path.join(T.REF_DIR, "libsvtref_pa.so"), mode=1)
    w, h = 328, 200
    luma = T.gen_clip(w, h, 1, 5)[0]
    planes, d = _alloc(w, h)
    assert T.oracle().svt_oracle_pa_prepare(luma.ctypes.data_as(C.c_void_p), luma.strides[0], C.byref(d), 1) == 0
    for s, p, mine in zip((1, 2, 4), PADS, planes):
        r = np.full_like(mine, 0xCD)
        dst = r.ctypes.data + p + p * r.strides[0]
        ref.eb_vp9_decimation_2d(C.c_void_p(luma.ctypes.data), luma.strides[0], w, h, C.c_void_p(dst), r.strides[0], s)
        ref.eb_vp9_generate_padding(C.c_void_p(r.ctypes.data), r.strides[0], w // s, h // s, p, p)
        assert np.array_equal(r, mine), s


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,quarter", [(328, 200, 1), (640, 360, 1), (3840, 2160, 0)])
def test_gpu_pa_vs_oracle(w, h, quarter):
    import torch
    lib = B.load()
    ctx = C.c_void_p()
    B.check(lib.svt_hip_ctx_create(C.byref(ctx), 0))
    try:
        dev = torch.device("cuda", 0)
        n = 2
        lumas = [T.gen_clip(w, h, 1, 7 + i)[0] for i in range(n)]
        want = []
        for luma in lumas:
            planes, d = _alloc(w, h)
            assert T.oracle().svt_oracle_pa_prepare(luma.ctypes.data_as(C.c_void_p), luma.strides[0], C.byref(d), quarter) == 0
            want.append(planes)
        d_l = [torch.from_numpy(l).to(dev) for l in lumas]
        d_p = [[torch.full(a.shape, 0xCD, dtype=torch.uint8, device=dev) for a in want[0]] for _ in range(n)]
        outs = (B.PaPicture * n)()
        for i in range(n):
            for name, t, p in zip(("full", "quarter", "sixteenth"), d_p[i], PADS):
                setattr(outs[i], name, B.plane_desc(want[0][("full", "quarter", "sixteenth").index(name)], p, p, ptr=t.data_ptr()))
        ptrs = (C.c_void_p * n)(*[t.data_ptr() for t in d_l])
        strides = (C.c_int32 * n)(*[w] * n)
        B.check(lib.svt_hip_pa_prepare_batch_device(ctx, n, ptrs, strides, outs, quarter))
        B.check(lib.svt_hip_ctx_synchronize(ctx))
        for i in range(n):
            for s in range(3):
                if s == 1 and not quarter:
                    continue
                assert np.array_equal(d_p[i][s].cpu().numpy(), want[i][s]), (i, s)
    finally:
        lib.svt_hip_ctx_destroy(ctx)
