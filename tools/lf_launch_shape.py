"""tools/lf_launch_shape.py -- the deblocking launch alone on the GPU, 2160p, 1 / 2 / 4 / 16 pictures per launch (ms per launch).  With
SVT_HIP_LF_ROWS / SVT_HIP_LF_EARLY for the launch-shape sweep of profiles/r05_lf_launch_shape.txt; SVT_HIP_LF_PROFILE=1 (+ SVT_HIP_LF_ROWTS=1)
prints the per-stage cycle shares (and per-row time stamps) of the kernel."""
import ctypes as C, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
torch.cuda.init()
import svt_testlib as T
B = T.B; lib = B.load()
ctx = C.c_void_p(); B.check(lib.svt_hip_ctx_create(C.byref(ctx), 0))
lib.svt_hip_last_kernel_ms.restype = C.c_float
case = T.make_lf_case(3, 3840, 2160)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
for N in (1, 2, 4, 16):
    keep = []
    d = (B.YuvPlanes * N)()
    lf = dev(case["lfm"].view(np.uint8)); keep.append(lf)
    for i in range(N):
        y, u, v = dev(case["y"]), dev(case["u"]), dev(case["v"]); keep += [y, u, v]
        d[i].y, d[i].u, d[i].v = y.data_ptr(), u.data_ptr(), v.data_ptr()
        d[i].y_stride, d[i].uv_stride, d[i].width, d[i].height = 3840, 1920, 3840, 2160
    ptrs = (C.c_void_p * N)(*([lf.data_ptr()] * N))
    ar = lambda x: (C.c_int32 * N)(*([x] * N))
    ts = []
    for i in range(6):
        B.check(lib.svt_hip_lf_batch_device(ctx, N, d, ptrs, ar(case["lfm"].shape[1]), C.byref(case["thr"]), ar(case["mi_rows"]), ar(case["mi_cols"]), 0))
        B.check(lib.svt_hip_ctx_synchronize(ctx))
        ts.append(lib.svt_hip_last_kernel_ms(ctx))
    print("N", N, "LF ms:", [round(t, 3) for t in ts], flush=True)
