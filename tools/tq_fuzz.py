"""Randomised parity sweep of the fused transform / quantise / reconstruct / distortion / rate pass against the oracle (GPU):
random plane sizes, quantiser steps from tiny (CAT6 levels, full blocks) to huge (empty blocks), extreme residuals, inter /
intra mixes (ADST types, row / column scans).  tools/tq_fuzz.py [cases] [seed]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
torch.cuda.init()   # before the product library: torch brings its own HIP runtime and must be the first to load one
import svt_testlib as T
B = T.B; lib = B.load()
ctx = C.c_void_p(); B.check(lib.svt_hip_ctx_create(C.byref(ctx), 0))
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for i in range(n_cases):
    w, h = 64 * int(rng.integers(1, 9)), 64 * int(rng.integers(1, 5))
    steps = tuple((int(a), int(a + rng.integers(0, 40))) for a in rng.choice([4, 8, 20, 40, 120, 400, 1336], 3))
    case = T.make_tq_case(int(rng.integers(1 << 20)), width=w, height=h, extreme=bool(rng.integers(0, 2)), qsteps=steps)
    rb = T.add_rate_info(case, int(rng.integers(1 << 20)), float(rng.random()))
    o, g = T.oracle_tq_rd_batch(case, rb), T.hip_tq_rd_batch_device(ctx, case)
    m = [n for n, a, b in zip(("recon", "qcoeff", "dqcoeff", "eob", "dist", "bits"), o, g) if not np.array_equal(a, b)]
    if m:
        bad += 1
        print("MISMATCH case", i, (w, h), steps, m)
print("cases", n_cases, "mismatches", bad)
sys.exit(1 if bad else 0)
