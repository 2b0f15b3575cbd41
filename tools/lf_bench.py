#!/usr/bin/env python3
"""tools/lf_bench.py [lib.so ...] -- deblocking kernel alone on the GPU: one 3840x2160 picture per launch (the latency of the
SB-row wavefront: what a temporal-layer wave of one GOP waits for) and 16 pictures per launch, device resident, averaged over
launches; the first launch of every variant is checked bit-exactly against the oracle.  Experiment aid (run through gpurun):
    python tools/lf_bench.py                       # the built library
    SVT_HIP_LF_PROFILE=1 python tools/lf_bench.py  # + the luma filter wave's cycle breakdown"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import svt_testlib as T
    B = T.B
    lib = B.load()
    W, H = int(os.environ.get("LF_W", 3840)), int(os.environ.get("LF_H", 2160))
    case = T.make_lf_case(3, W, H)
    want = T.oracle_lf_frame(case)
    dev = torch.device("cuda", 0)
    ctx = C.c_void_p()
    B.check(lib.svt_hip_ctx_create(C.byref(ctx), 0))
    lfm = torch.from_numpy(np.ascontiguousarray(case["lfm"]).view(np.uint8)).to(dev)
    sb_cols = case["lfm"].shape[1]
    for n_pics in (1, 16):
        src = [torch.from_numpy(np.ascontiguousarray(case[k])).to(dev) for k in "yuv"]
        bufs = [[t.clone() for t in src] for _ in range(n_pics)]
        descs = (B.YuvPlanes * n_pics)()
        for k, (y, u, v) in enumerate(bufs):
            d = descs[k]
            d.y, d.u, d.v, d.y_stride, d.uv_stride, d.width, d.height = y.data_ptr(), u.data_ptr(), v.data_ptr(), W, W // 2, W, H
        lfms = (C.c_void_p * n_pics)(*[lfm.data_ptr()] * n_pics)
        i32 = lambda v: (C.c_int32 * n_pics)(*[v] * n_pics)
        args = (ctx, n_pics, descs, lfms, i32(sb_cols), C.byref(case["thr"]), i32(case["mi_rows"]), i32(case["mi_cols"]), 0)
        torch.cuda.synchronize()
        B.check(lib.svt_hip_lf_batch_device(*args))
        B.check(lib.svt_hip_ctx_synchronize(ctx))
        ok = all(np.array_equal(t.cpu().numpy(), w) for t, w in zip(bufs[n_pics - 1], want))
        ms = []
        for _ in range(8):      # filtering filtered pictures again: same work, different data
            for k in range(n_pics):
                for t, s in zip(bufs[k], src):
                    t.copy_(s)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            B.check(lib.svt_hip_lf_batch_device(*args))
            B.check(lib.svt_hip_ctx_synchronize(ctx))
            ms.append((time.perf_counter() - t0) * 1e3)
        print(f"lf {W}x{H} pics={n_pics:2d}  bit-exact={ok}  ms/launch min {min(ms):.3f} median {sorted(ms)[len(ms) // 2]:.3f}  kernel_ms(last) {lib.svt_hip_last_kernel_ms(ctx):.3f}",
              flush=True)
    lib.svt_hip_ctx_destroy(ctx)


if __name__ == "__main__":
    main()
