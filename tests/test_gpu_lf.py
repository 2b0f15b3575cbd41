"""GPU parity: HIP in-loop deblocking of whole frames (through the C ABI) vs the oracle, bit-exact."""
import ctypes as C

import numpy as np
import pytest

import svt_testlib as T

B = T.B
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    lib = B.load()
    c = C.c_void_p()
    B.check(lib.svt_hip_ctx_create(C.byref(c), 0))
    yield c
    lib.svt_hip_ctx_destroy(c)


@pytest.mark.parametrize("w,h,seed", [(192, 128, 1), (200, 136, 2), (328, 200, 3), (64, 64, 4), (72, 72, 5), (648, 360, 7), (72, 136, 8)])
@pytest.mark.parametrize("sharp", [0, 4])
def test_lf_frame_vs_oracle(ctx, w, h, seed, sharp):
    case = T.make_lf_case(seed, w, h, sharp)
    o = T.oracle_lf_frame(case)
    g = T.hip_lf_frame(ctx, case)
    for n, a, b in zip("yuv", o, g):
        assert np.array_equal(a, b), (n, int(np.sum(a != b)), np.argwhere(a != b)[:6].tolist())
    assert np.mean(o[0] != case["y"]) > 0.02  # the filter really did something


def test_lf_y_only_and_repeatability(ctx):
    case = T.make_lf_case(9, 1280, 720)
    o = T.oracle_lf_frame(case, y_only=True)
    g = T.hip_lf_frame(ctx, case, y_only=True)
    assert np.array_equal(o[0], g[0]) and np.array_equal(g[1], case["u"]) and np.array_equal(g[2], case["v"])
    # the SB wavefront must give the same answer every time (ordering bugs show up as run-to-run differences)
    full = T.oracle_lf_frame(case)
    for _ in range(5):
        g = T.hip_lf_frame(ctx, case)
        assert all(np.array_equal(a, b) for a, b in zip(full, g))


@pytest.mark.parametrize("sizes", [((328, 200), (136, 64), (256, 192)),
                                   ((328, 200), (136, 64), (256, 192), (640, 360), (72, 72), (200, 136), (648, 264))])
def test_lf_batch_of_different_pictures(ctx, sizes):
    """svt_hip_lf_batch_device: pictures of different sizes (1 .. 6 SB rows; partial SBs) in one launch -- the persistent
    workgroups interleave their rows by ticket; every picture must come out as if filtered alone.  Three pictures run on the latency
    instance of the kernel (every SB row resident, seam rows handed to the row below early), seven on the throughput instance."""
    import torch
    lib = B.load()
    cases = [T.make_lf_case(21 + k, w, h) for k, (w, h) in enumerate(sizes)]
    n = len(cases)
    keep, descs = [], (B.YuvPlanes * n)()

    def dev(a):
        t = torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1)).cuda()
        keep.append(t)
        return t

    planes, lfms = [], []
    for k, c in enumerate(cases):
        y, u, v = dev(c["y"]), dev(c["u"]), dev(c["v"])
        planes.append((y, u, v))
        lfms.append(dev(c["lfm"]))
        d = descs[k]
        d.y, d.u, d.v = y.data_ptr(), u.data_ptr(), v.data_ptr()
        d.y_stride, d.uv_stride, d.width, d.height = c["y"].shape[1], c["u"].shape[1], c["y"].shape[1], c["y"].shape[0]
    ptrs = (C.c_void_p * n)(*[t.data_ptr() for t in lfms])
    arr = lambda vals: (C.c_int32 * n)(*vals)
    B.check(lib.svt_hip_lf_batch_device(ctx, n, descs, ptrs, arr([c["lfm"].shape[1] for c in cases]), C.byref(cases[0]["thr"]),
                                        arr([c["mi_rows"] for c in cases]), arr([c["mi_cols"] for c in cases]), 0))
    B.check(lib.svt_hip_ctx_synchronize(ctx))
    for c, (y, u, v) in zip(cases, planes):
        oy, ou, ov = T.oracle_lf_frame(c)
        assert np.array_equal(y.cpu().numpy().reshape(oy.shape), oy)
        assert np.array_equal(u.cpu().numpy().reshape(ou.shape), ou) and np.array_equal(v.cpu().numpy().reshape(ov.shape), ov)
