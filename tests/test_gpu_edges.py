"""GPU parity on the smallest and the widest pictures the reference accepts (W, H in [64, 8192] x [64, 4320] for the
encoder, Codec/EbEncHandle.c:2295-2340; the per-stage entry points go down to one 8x8 unit): single-SB pictures,
partial SBs in both directions, one-SB-row pictures of maximum width."""
import ctypes as C

import numpy as np
import pytest

import me_configs as MC
import svt_testlib as T
from test_gpu_me import hip_me_picture

B = T.B
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    lib = B.load()
    c = C.c_void_p()
    B.check(lib.svt_hip_ctx_create(C.byref(c), 0))
    yield c
    lib.svt_hip_ctx_destroy(c)


@pytest.mark.parametrize("w,h", [(64, 64), (72, 80), (136, 64), (64, 200), (8192, 64)])
def test_me_extreme_shapes(ctx, w, h):
    pics = [T.PaPic(f) for f in T.gen_clip_subpel(w, h, 3, 3)]
    for name in MC.PRESETS:
        p = MC.preset(name, 2, 1)
        o, _ = T.oracle_me_picture(pics[1], pics[0], pics[2], p)
        g, _ = hip_me_picture(ctx, pics[1], pics[0], pics[2], p)
        assert not T.me_results_equal(o, g, 2), name


@pytest.mark.parametrize("w,h", [(64, 64), (8, 8), (16, 72), (72, 16), (4096, 64)])
def test_lf_extreme_shapes(ctx, w, h):
    case = T.make_lf_case(4, w, h)
    for a, b in zip(T.hip_lf_frame(ctx, case), T.oracle_lf_frame(case)):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("w,h", [(8, 8), (16, 8), (64, 64), (8, 72)])
def test_mc_extreme_shapes(ctx, w, h):
    case = T.make_mc_case(4, width=w, height=h)
    for a, b in zip(T.hip_mc_frame(ctx, case), T.oracle_mc_frame(case)):
        assert np.array_equal(a, b)


def test_device_set_and_reference_handoff_single_device():
    """one-process multi-GPU host side on the one GPU there is: a device set of two contexts on device 0 and the hand-off of a
    padded reference buffer between them (producer writes, consumer reads after the hand-off, producer overwrites afterwards)"""
    import torch
    lib = B.load()
    lib.svt_hip_device_set_ctx.restype = C.c_void_p
    st = C.c_void_p()
    B.check(lib.svt_hip_device_set_create(C.byref(st), (C.c_int32 * 2)(0, 0), 2))
    try:
        assert lib.svt_hip_device_set_size(st) == 2
        a, b = C.c_void_p(lib.svt_hip_device_set_ctx(st, 0)), C.c_void_p(lib.svt_hip_device_set_ctx(st, 1))
        assert a.value and b.value and a.value != b.value and not lib.svt_hip_device_set_ctx(st, 2)
        n = (3840 + 160) * (2160 + 160) * 3 // 2     # a padded 4K reference picture: 13.9 MB
        src, dst = C.c_void_p(), C.c_void_p()
        B.check(lib.svt_hip_mem_alloc(a, n, C.byref(src))); B.check(lib.svt_hip_mem_alloc(b, n, C.byref(dst)))
        for value in (0x5A, 0xC3):
            B.check(lib.svt_hip_mem_set(a, src, value, n))                        # producer's stream
            B.check(lib.svt_hip_ref_handoff_device(a, src, b, dst, n))            # ordered behind it, onto the consumer's stream
            B.check(lib.svt_hip_mem_set(a, src, 0, n))                            # the producer reuses its buffer at once
            host = np.zeros(n, np.uint8)
            B.check(lib.svt_hip_mem_download(b, host.ctypes.data_as(C.c_void_p), dst, n))
            assert (host == value).all()
        lib.svt_hip_mem_free(a, src); lib.svt_hip_mem_free(b, dst)
        assert lib.svt_hip_ref_handoff_device(a, None, b, dst, n) != 0
    finally:
        lib.svt_hip_device_set_destroy(st)
