"""Randomised parity sweep of the deblocking kernel against the oracle (GPU): random sizes (multiples of 8, partial SBs), mask /
level / content seeds, sharpness 0..7.  tools/lf_fuzz.py [cases] [seed]"""
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
    w, h = 8 * int(rng.integers(8, 60)), 8 * int(rng.integers(8, 40))
    case = T.make_lf_case(int(rng.integers(1 << 20)), w, h, int(rng.integers(0, 8)))
    if rng.integers(0, 3) == 0:   # smooth content: the 8- and 16-wide (flat) filters take part
        ramp = (np.add.outer(np.arange(h), np.arange(w)) // 6 % 200 + 20).astype(np.uint8)
        case["y"][:] = ramp + rng.integers(0, 2, ramp.shape, dtype=np.uint8)
        case["u"][:] = ramp[::2, ::2]; case["v"][:] = 255 - ramp[::2, ::2]
    o, g = T.oracle_lf_frame(case), T.hip_lf_frame(ctx, case)
    if not all(np.array_equal(a, b) for a, b in zip(o, g)):
        bad += 1
        print("MISMATCH case", i, (w, h))
print("cases", n_cases, "mismatches", bad)
sys.exit(1 if bad else 0)
