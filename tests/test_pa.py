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
@pytest.mark.parametrize("w,h,quarter", [(328, 200, 1), (640, 360, 1), (3840, 2160, 0), (8, 8, 1), (24, 16, 1), (72, 64, 1), (136, 72, 0),
                                         (1000, 568, 1), (1096, 600, 0)])
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


def _numpy_meanvar(padded, pad, w, h):
    nx, ny = (w + 63) // 64, (h + 63) // 64
    mean, var = np.zeros((nx * ny, 85), np.uint8), np.zeros((nx * ny, 85), np.uint16)
    for sb in range(nx * ny):
        y0, x0 = pad + (sb // nx) * 64, pad + (sb % nx) * 64
        blk = padded[y0:y0 + 64:2, x0:x0 + 64].astype(np.uint64)          # rows 0,2,4,.. of the SB
        s8 = blk.reshape(8, 4, 8, 8).sum(axis=(1, 3)) << np.uint64(3)     # [by][bx]
        q8 = (blk * blk).reshape(8, 4, 8, 8).sum(axis=(1, 3)) << np.uint64(11)
        def up(a):
            n = a.shape[0] // 2
            return a.reshape(n, 2, n, 2).sum(axis=(1, 3)) >> np.uint64(2)
        m = {8: s8, 16: up(s8)}; q = {8: q8, 16: up(q8)}
        m[32], q[32] = up(m[16]), up(q[16]); m[64], q[64] = up(m[32]), up(q[32])
        mm = np.concatenate([m[64].ravel(), m[32].ravel(), m[16].ravel(), m[8].ravel()])
        qq = np.concatenate([q[64].ravel(), q[32].ravel(), q[16].ravel(), q[8].ravel()])
        mean[sb] = (mm >> np.uint64(8)).astype(np.uint8)
        var[sb] = ((qq - mm * mm) >> np.uint64(16)).astype(np.uint16)
    return mean, var


def _oracle_meanvar(pic, w, h):
    n = T.n_sb(w, h)
    mean, var = np.zeros((n, 85), np.uint8), np.zeros((n, 85), np.uint16)
    d = pic.desc()
    assert T.oracle().svt_oracle_pa_mean_variance(C.byref(d.full), mean.ctypes.data_as(C.c_void_p), var.ctypes.data_as(C.c_void_p)) == 0
    return mean, var


@pytest.mark.parametrize("w,h", [(128, 64), (328, 200)])
def test_oracle_meanvar_vs_numpy(w, h):
    pic = T.PaPic(T.gen_clip(w, h, 1, 9)[0])
    om, ov = _oracle_meanvar(pic, w, h)
    nm, nv = _numpy_meanvar(pic.full, 68, w, h)
    assert np.array_equal(om, nm) and np.array_equal(ov, nv)
    assert ov.max() > 100


@pytest.mark.skipif(not os.path.exists(os.path.join(T.REF_DIR, "libsvtref_pa.so")), reason="oracle/_ref/libsvtref_pa.so not built")
def test_oracle_mean8x8_vs_reference_sse2_leaf():
    ref = C.CDLL(os.path.join(T.REF_DIR, "libsvtref_pa.so"), mode=1)
    ref.eb_vp9_compute_sub_mean8x8_sse2_intrin.restype = C.c_uint64
    ref.eb_vp9_compute_subd_mean_of_squared_values8x8_sse2_intrin.restype = C.c_uint64
    rng = np.random.default_rng(2)
    for _ in range(50):
        a = rng.integers(0, 256, (8, 24), dtype=np.uint8)
        if rng.integers(0, 4) == 0:
            a[:] = 255
        m, q = C.c_uint64(), C.c_uint64()
        T.oracle().svt_oracle_pa_mean8x8(C.c_void_p(a.ctypes.data + 5), a.strides[0], C.byref(m), C.byref(q))
        assert m.value == ref.eb_vp9_compute_sub_mean8x8_sse2_intrin(C.c_void_p(a.ctypes.data + 5), C.c_uint16(a.strides[0]))
        assert q.value == ref.eb_vp9_compute_subd_mean_of_squared_values8x8_sse2_intrin(C.c_void_p(a.ctypes.data + 5), C.c_uint16(a.strides[0]))


@pytest.mark.gpu
@pytest.mark.parametrize("w,h", [(328, 200), (3840, 2160)])
def test_gpu_meanvar_vs_oracle(w, h):
    import torch
    lib = B.load()
    ctx = C.c_void_p()
    B.check(lib.svt_hip_ctx_create(C.byref(ctx), 0))
    try:
        dev = torch.device("cuda", 0)
        pic = T.PaPic(T.gen_clip(w, h, 1, 12)[0])
        om, ov = _oracle_meanvar(pic, w, h)
        t = torch.from_numpy(pic.full).to(dev)
        n = T.n_sb(w, h)
        dm, dv = torch.zeros(n * 85, dtype=torch.uint8, device=dev), torch.zeros(n * 85, dtype=torch.int16, device=dev)
        pl = B.plane_desc(pic.full, 68, 68, ptr=t.data_ptr())
        B.check(lib.svt_hip_pa_mean_variance_device(ctx, C.byref(pl), C.c_void_p(dm.data_ptr()), C.c_void_p(dv.data_ptr())))
        B.check(lib.svt_hip_ctx_synchronize(ctx))
        assert np.array_equal(dm.cpu().numpy().reshape(n, 85), om)
        assert np.array_equal(dv.cpu().numpy().view(np.uint16).reshape(n, 85), ov)
    finally:
        lib.svt_hip_ctx_destroy(ctx)
