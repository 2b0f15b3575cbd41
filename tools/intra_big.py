"""Large-picture parity runs of the intra pass (up to 3840x2160: thousands of areas in flight, every hand-over between workgroups on
different CUs / XCDs) against the oracle chain.  `python tools/intra_big.py [n_cases] [seed]` on the GPU box."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np

import svt_testlib as T
import test_gpu_intra as TI

B = T.B
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = C.c_void_p()
B.check(B.load().svt_hip_ctx_create(C.byref(ctx), 0))
for k in range(n_cases):
    W, H = 8 * int(rng.integers(100, 481)), 8 * int(rng.integers(60, 271))
    sizes = [(8, 16, 32), (4, 8, 16, 32), (8,), (4, 8), (8, 32)][int(rng.integers(0, 5))]
    q = int(rng.integers(20, 230))
    TI.check(ctx, W, H, 5000 + k, q, TI.KEY, sizes=sizes, quality=False)
    print("ok", k, W, H, sizes, q, flush=True)
print(f"intra big: {n_cases} cases OK", flush=True)
