"""Randomised parity sweep of the inter-prediction kernel against the oracle (GPU): random sizes (multiples of 8), motion-vector
ranges up to far outside the picture, both use_subpel modes, rectangular blocks, intra shares.  tools/mc_fuzz.py [cases] [seed]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import svt_testlib as T
B = T.B; lib = B.load()
ctx = C.c_void_p(); B.check(lib.svt_hip_ctx_create(C.byref(ctx), 0))
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for i in range(n_cases):
    w, h = 8 * int(rng.integers(8, 80)), 8 * int(rng.integers(8, 48))
    case = T.make_mc_case(int(rng.integers(1 << 20)), width=w, height=h, mv_range=int(rng.choice([4, 16, 48, 200, 600])),
                          use_subpel=int(rng.integers(0, 2)), rect=bool(rng.integers(0, 2)), intra_share=float(rng.choice([0.0, 0.1, 0.5])))
    o, g = T.oracle_mc_frame(case), T.hip_mc_frame(ctx, case)
    if not all(np.array_equal(a, b) for a, b in zip(o, g)):
        bad += 1
        print("MISMATCH case", i, (w, h))
print("cases", n_cases, "mismatches", bad)
sys.exit(1 if bad else 0)
