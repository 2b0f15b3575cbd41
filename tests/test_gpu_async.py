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
