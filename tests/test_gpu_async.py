"""C-ABI plumbing a non-blocking host needs (GPU): svt_hip_mem_upload_2d_async copies the caller's rows before it returns (the
caller may scribble over them at once) and more uploads than the context has staging buffers still arrive intact and in order;
svt_hip_ctx_marker_record / _query / _wait follow the stream: a marker recorded behind work completes after it, query never blocks,
markers older than the ring are reported complete."""
import ctypes as C

import numpy as np
import pytest

import svt_testlib as T

B = T.B
pytestmark = pytest.mark.gpu


def test_async_upload_and_markers():
    import torch
    lib = B.load()
    lib.svt_hip_ctx_marker_query.restype = C.c_int32
    ctx = C.c_void_p()
    B.check(lib.svt_hip_ctx_create(C.byref(ctx), 0))
    try:
        W, H, n = 1920, 1080, 12                       # 12 uploads through 4 staging buffers
        rng = np.random.default_rng(1)
        dst = [torch.zeros((H, W + 64), dtype=torch.uint8, device="cuda") for _ in range(n)]
        torch.cuda.synchronize()
        want, markers = [], []
        buf = np.zeros((H, W + 32), np.uint8)          # source rows with their own stride
        for k in range(n):
            buf[:, :W] = rng.integers(0, 256, (H, W), dtype=np.uint8)
            want.append(buf[:, :W].copy())
            B.check(lib.svt_hip_mem_upload_2d_async(ctx, C.c_void_p(dst[k].data_ptr()), C.c_size_t(W + 64), buf.ctypes.data_as(C.c_void_p), C.c_size_t(W + 32),
                                                    C.c_size_t(W), C.c_size_t(H)))
            buf[:] = 0xEE                              # the call has copied the rows: the caller's buffer is its own again
            m = C.c_uint64()
            B.check(lib.svt_hip_ctx_marker_record(ctx, C.byref(m)))
            markers.append(m.value)
        assert markers == list(range(markers[0], markers[0] + n))
        assert lib.svt_hip_ctx_marker_query(ctx, C.c_uint64(markers[-1])) in (0, 1)      # never blocks, never fails
        B.check(lib.svt_hip_ctx_marker_wait(ctx, C.c_uint64(markers[-1])))
        assert all(lib.svt_hip_ctx_marker_query(ctx, C.c_uint64(m)) == 1 for m in markers)   # stream order: everything before it too
        for k in range(n):
            got = dst[k].cpu().numpy()
            assert np.array_equal(got[:, :W], want[k]) and not got[:, W:].any(), k
        assert lib.svt_hip_ctx_marker_query(ctx, C.c_uint64(markers[-1] + 5)) < 0           # a marker never handed out
        for _ in range(1100):                           # wrap the marker ring: old markers stay "complete"
            m = C.c_uint64()
            B.check(lib.svt_hip_ctx_marker_record(ctx, C.byref(m)))
        assert lib.svt_hip_ctx_marker_query(ctx, C.c_uint64(markers[0])) == 1
        B.check(lib.svt_hip_ctx_marker_wait(ctx, C.c_uint64(markers[0])))
    finally:
        lib.svt_hip_ctx_destroy(ctx)


def test_planes_upload_one_slot_one_copy():
    """svt_hip_mem_upload_planes_async: the planes of a picture through one staging slot -- back-to-back tight destinations (Y | Cb | Cr in one
    buffer: one host-to-device copy) and separate, strided destinations; the caller's rows may be overwritten as soon as the call returns,
    and more pictures than staging slots arrive intact."""
    import torch
    lib = B.load()
    ctx = C.c_void_p()
    B.check(lib.svt_hip_ctx_create(C.byref(ctx), 0))
    try:
        W, H, n = 640, 360, 7
        rng = np.random.default_rng(9)
        P, S = C.c_void_p * 3, C.c_size_t * 3
        wb, rows = (W, W // 2, W // 2), (H, H // 2, H // 2)
        src_st = (W + 16, W // 2 + 8, W // 2)
        tight = [torch.zeros(W * H * 3 // 2, dtype=torch.uint8, device="cuda") for _ in range(n)]
        apart = [[torch.zeros((r, w + 32), dtype=torch.uint8, device="cuda") for w, r in zip(wb, rows)] for _ in range(n)]
        torch.cuda.synchronize()
        bufs = [np.zeros((r, st), np.uint8) for r, st in zip(rows, src_st)]
        want = []
        for k in range(n):
            for b_, w in zip(bufs, wb):
                b_[:, :w] = rng.integers(0, 256, (b_.shape[0], w), dtype=np.uint8)
            want.append([b_[:, :w].copy() for b_, w in zip(bufs, wb)])
            base = tight[k].data_ptr()
            B.check(lib.svt_hip_mem_upload_planes_async(ctx, 3, P(base, base + W * H, base + W * H + W * H // 4), S(*wb), P(*[b_.ctypes.data for b_ in bufs]), S(*src_st),
                                                        S(*wb), S(*rows)))
            B.check(lib.svt_hip_mem_upload_planes_async(ctx, 3, P(*[t.data_ptr() for t in apart[k]]), S(*[w + 32 for w in wb]), P(*[b_.ctypes.data for b_ in bufs]), S(*src_st),
                                                        S(*wb), S(*rows)))
            for b_ in bufs:
                b_[:] = 0xEE                             # the caller reuses its rows at once
        B.check(lib.svt_hip_ctx_synchronize(ctx))
        for k in range(n):
            flat = tight[k].cpu().numpy()
            off = 0
            for j, (w, r) in enumerate(zip(wb, rows)):
                assert np.array_equal(flat[off:off + w * r].reshape(r, w), want[k][j]), (k, j)
                off += w * r
                a = apart[k][j].cpu().numpy()
                assert np.array_equal(a[:, :w], want[k][j]) and not a[:, w:].any(), (k, j)
        # argument checks: a null plane, a destination stride below the width
        assert lib.svt_hip_mem_upload_planes_async(ctx, 3, P(tight[0].data_ptr(), None, None), S(*wb), P(*[b_.ctypes.data for b_ in bufs]), S(*src_st), S(*wb), S(*rows)) != 0
        assert lib.svt_hip_mem_upload_planes_async(ctx, 1, P(tight[0].data_ptr(), None, None), S(W - 1, 0, 0), P(bufs[0].ctypes.data, None, None), S(*src_st), S(*wb), S(*rows)) != 0
    finally:
        lib.svt_hip_ctx_destroy(ctx)


def test_context_warm_up_and_reservations():
    """what a host calls before it starts a clock: svt_hip_ctx_warm / _warm_scratch submit an empty kernel (with a private
    segment of the asked size) and wait; svt_hip_lf_reserve takes the deblocking launches' descriptor buffer at its largest -- a launch behind it
    gives the same pictures as one that grows the buffer on demand; bad arguments are refused; a staging request larger than a ring entry's
    share of the slab still works (the entry grows on its own)."""
    lib = B.load()
    for f in ("svt_hip_ctx_warm", "svt_hip_ctx_warm_scratch", "svt_hip_lf_reserve"):
        getattr(lib, f).restype = C.c_int32
    a, b = C.c_void_p(), C.c_void_p()
    B.check(lib.svt_hip_ctx_create(C.byref(a), 0))
    B.check(lib.svt_hip_ctx_create(C.byref(b), 0))
    try:
        assert lib.svt_hip_ctx_warm(a) == 0
        for nbytes in (0, 130, 256, 752, 1024):
            assert lib.svt_hip_ctx_warm_scratch(a, nbytes) == 0
        assert lib.svt_hip_ctx_warm(None) == -1 and lib.svt_hip_ctx_warm_scratch(a, -1) == -1
        assert lib.svt_hip_lf_reserve(a, 4, 136 // 8, 200 // 8) == 0
        assert lib.svt_hip_lf_reserve(a, 0, 17, 25) == -1 and lib.svt_hip_lf_reserve(None, 1, 17, 25) == -1 and lib.svt_hip_lf_reserve(a, 1, 0, 25) == -1
        case = T.make_lf_case(7, 200, 136)
        want = T.oracle_lf_frame(case)
        for ctx in (a, b):   # reserved / grown on demand
            for name, o, g in zip("yuv", want, T.hip_lf_frame(ctx, case)):
                assert np.array_equal(o, g), name
        big = T.make_lf_case(9, 1920, 1088)   # descriptors of 510 SBs: more than the reservation of context a -> the buffer grows
        for name, o, g in zip("yuv", T.oracle_lf_frame(big), T.hip_lf_frame(a, big)):
            assert np.array_equal(o, g), name
    finally:
        lib.svt_hip_ctx_destroy(a)
        lib.svt_hip_ctx_destroy(b)
