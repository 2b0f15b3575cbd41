"""Randomised parity runs of the intra encode pass on the GPU against the oracle chain: picture sizes (multiples of 8, any SB
remainder), q indices, block-size sets, mode subsets, flag combinations.  `python tools/intra_fuzz.py [n_cases] [seed]` on the GPU box."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np

import svt_testlib as T
import test_gpu_intra as TI

B = T.B
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = C.c_void_p()
B.check(B.load().svt_hip_ctx_create(C.byref(ctx), 0))
for k in range(n_cases):
    W, H = 8 * int(rng.integers(2, 60)), 8 * int(rng.integers(2, 40))
    q = int(rng.integers(1, 256))
    sizes = [(8,), (8, 16), (8, 32), (8, 16, 32), (4,), (4, 8), (4, 8, 16, 32), (4, 16)][int(rng.integers(0, 8))]
    modes = tuple(sorted(set(int(m) for m in rng.integers(0, 10, int(rng.integers(1, 6))))))
    cfg = dict(enc_mode=int(rng.integers(0, 10)), tune=int(rng.integers(0, 3)), temporal_layer_index=0, is_used_as_reference=1, recon_file=int(rng.integers(0, 2)),
               loop_filter=int(rng.integers(0, 2)))
    try:
        TI.check(ctx, W, H, 1000 + k, q, cfg, sizes=sizes, modes=modes, quality=False)
    except AssertionError as e:
        print("FAIL", k, W, H, q, sizes, modes, cfg, str(e)[:200], flush=True)
        sys.exit(1)
print(f"intra fuzz: {n_cases} cases OK", flush=True)
