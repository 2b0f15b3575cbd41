import sys, time, ctypes as C
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, torch
import svt_testlib as T, encdec_model as M
from test_gpu_encdec import flags_of
import test_gpu_intra as TI
B = T.B
lib = B.load()
c = C.c_void_p(); B.check(lib.svt_hip_ctx_create(C.byref(c), 0))
W, H = 3840, 2160
src = T.gen_yuv(W, H, 11)
thr = B.LfThresh(); lib.svt_hip_lf_thresh_init(C.byref(thr), 0)
flags = flags_of(**TI.KEY)
for sizes in ((8, 16, 32), (8, 32), (8, 16), (8,)):
    mi = M.gen_intra_grid(11, W, H, sizes=sizes) if len(sizes) == 3 else M.gen_intra_grid(11, W, H, sizes=sizes)
    if len(sizes) == 2:   # force the large size everywhere it fits
        import numpy.random
        class R:  # rng stub: always choose the big block
            def random(self): return 0.0
            def choice(self, m): return 0
        # simple uniform grid
        n8 = sizes[1] // 8
        mi = np.zeros((H // 8, W // 8), dtype=B.LF_MODE_INFO_DTYPE)
        for r in range(0, H // 8, n8):
            for cc in range(0, W // 8, n8):
                if r + n8 <= H // 8 and cc + n8 <= W // 8:
                    mi[r:r+n8, cc:cc+n8]["sb_type"] = {2: 6, 4: 9}[n8]; mi[r:r+n8, cc:cc+n8]["tx_size"] = {2: 2, 4: 3}[n8]
                else:
                    mi[r:r+n8, cc:cc+n8]["sb_type"] = 3; mi[r:r+n8, cc:cc+n8]["tx_size"] = 1
        mi["filter_level"] = 20
        mi["pad"][..., 1] = 9; mi["pad"][..., 2] = 4
    t = []
    for k in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        g = TI.run_intra(c, src, mi, 140, flags, thr, M.RefPic(W, H), want_pred=False)
        t.append(time.perf_counter() - t0)
    print(sizes, "rc", g["rc"], "wall incl. upload/download %.1f ms" % (1e3 * min(t)), flush=True)
