"""Times the intra encode pass (svt_hip_encdec_intra_device, deblocking and border included) of one 2160p picture for uniform block
sizes and for a random partition: wall clock around call + stream synchronisation, inputs resident.  Run on the GPU box."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import torch

import encdec_model as M
import svt_testlib as T
import test_gpu_intra as TI
from test_gpu_encdec import dev, flags_of

B = T.B
lib = B.load()
ctx = C.c_void_p()
B.check(lib.svt_hip_ctx_create(C.byref(ctx), 0))
W, H = 3840, 2160
src = T.gen_yuv(W, H, 11)
thr = B.LfThresh()
lib.svt_hip_lf_thresh_init(C.byref(thr), 0)
flags = flags_of(**TI.KEY)


def uniform(n8, ymode, uvmode):
    mi = np.zeros((H // 8, W // 8), dtype=B.LF_MODE_INFO_DTYPE)
    r, c = np.meshgrid(np.arange(H // 8), np.arange(W // 8), indexing="ij")
    fit = ((r // n8) * n8 + n8 <= H // 8) & ((c // n8) * n8 + n8 <= W // 8)
    mi["sb_type"], mi["tx_size"] = np.where(fit, {1: 3, 2: 6, 4: 9}[n8], 3), np.where(fit, {1: 1, 2: 2, 4: 3}[n8], 1)
    mi["filter_level"] = 20
    mi["pad"][..., 1], mi["pad"][..., 2] = ymode, uvmode
    return mi


srcb = dev(np.concatenate([p.ravel() for p in src]))
nco = T.n_sb(W, H) * B.SB_COEFFS
q_t, dq_t = torch.zeros(nco, dtype=torch.int16, device="cuda"), torch.zeros(nco, dtype=torch.int16, device="cuda")
rec = M.RefPic(W, H)
rec_t = dev(rec.buf)
emap_t = torch.zeros(M.eob_map_offsets(W, H)[3], dtype=torch.int16, device="cuda")
lfm_t = torch.zeros(T.n_sb(W, H) * 160, dtype=torch.uint8, device="cuda")
nz_t = torch.zeros(W * H // 64, dtype=torch.uint8, device="cuda")
work = C.c_void_p()
B.check(lib.svt_hip_encdec_work_create(ctx, 1, W, H, C.byref(work)))
for name, mi in (("32x32 TM", uniform(4, 9, 9)), ("16x16 DC (the stand-in)", uniform(2, 0, 0)), ("16x16 D45", uniform(2, 3, 3)), ("8x8 TM", uniform(1, 9, 9)),
                 ("random 8..32, all modes", M.gen_intra_grid(11, W, H))):
    lf_t = dev(np.ascontiguousarray(mi).view(np.uint8))
    p = B.EncdecPicture()
    p.d_lf_mi = lf_t.data_ptr()
    d = B.YuvPlanes()
    base = srcb.data_ptr()
    d.y, d.u, d.v, d.y_stride, d.uv_stride, d.width, d.height = base, base + W * H, base + W * H + (W // 2) * (H // 2), W, W // 2, W, H
    p.src = d
    p.recon = rec.desc(rec_t.data_ptr())
    p.d_qcoeff, p.d_dqcoeff, p.d_eob_map, p.d_lfm, p.d_nz = q_t.data_ptr(), dq_t.data_ptr(), emap_t.data_ptr(), lfm_t.data_ptr(), nz_t.data_ptr()
    torch.cuda.synchronize()
    ts = []
    for k in range(4):
        t0 = time.perf_counter()
        B.check(lib.svt_hip_encdec_intra_device(ctx, work, C.byref(p), W, H, W // 8, 140, C.byref(flags), C.byref(thr), M.PAD, M.PAD))
        assert lib.svt_hip_encdec_work_status(ctx, work, None) == 0
        ts.append(time.perf_counter() - t0)
    print("%-28s %.2f ms" % (name, 1e3 * min(ts)), flush=True)
lib.svt_hip_encdec_work_destroy(ctx, work)
