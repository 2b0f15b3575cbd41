"""Randomised parity sweep of the ME kernel against the oracle (GPU): random picture sizes (incl. partial SBs), content kinds
(smooth motion, noise, flat / tie-heavy, blocky), presets, list counts and temporal layers.  tools/me_fuzz.py [cases] [seed] [fast]
`fast`: the 2160p enc-mode-8 preset on pictures of whole SB columns only (csrc/me_fast.h's driver; the run checks that it served).
`c5`: the 2160p enc-mode-3 preset (64x64 area, SSD search, three HME levels) with mutated search areas (multiples of 8: the compact LDS layout
with its second launch, csrc/me_layout.h; others: the plain one), 8x8 modes and metrics -- on small pictures most SBs meet clipped areas."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import me_configs as MC, svt_testlib as T
B = T.B; lib = B.load()
ctx = C.c_void_p(); B.check(lib.svt_hip_ctx_create(C.byref(ctx), 0))
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
fast = len(sys.argv) > 3 and sys.argv[3] == "fast"
c5 = len(sys.argv) > 3 and sys.argv[3] == "c5"
lib.svt_hip_me_last_instance.argtypes = [C.c_void_p]

def content(kind, w, h):
    if kind == 0: return list(T.gen_clip(w, h, 3, int(rng.integers(1 << 20))))
    if kind == 1: return list(T.gen_clip_subpel(w, h, 3, int(rng.integers(1 << 20))))
    if kind == 2: return [rng.integers(0, 256, (h, w), dtype=np.uint8) for _ in range(3)]
    if kind == 3:
        v = int(rng.integers(0, 250)); return [np.full((h, w), v + d, np.uint8) for d in (0, 0, int(rng.integers(0, 3)))]
    base = np.kron(rng.integers(0, 256, (h // 8 + 1, w // 8 + 1)), np.ones((8, 8)))[:h, :w].astype(np.uint8)   # blocky: many ties
    return [np.roll(base, (int(rng.integers(-3, 4)), int(rng.integers(-5, 6))), (0, 1)) for _ in range(3)]

def hip(cur, r0, r1, p):
    nsb = T.n_sb(cur.luma.shape[1], cur.luma.shape[0])
    res = np.zeros((nsb, 85), dtype=B.ME_RESULT_DTYPE); rc = np.zeros(nsb, np.uint32)
    dc, d0 = cur.desc(), r0.desc(); d1 = r1.desc() if r1 is not None else None
    B.check(lib.svt_hip_me_picture(ctx, C.byref(dc), C.byref(d0), C.byref(d1) if d1 is not None else None, C.byref(p),
                                   res.ctypes.data_as(C.c_void_p), rc.ctypes.data_as(C.c_void_p)))
    return res

bad = 0
names = list(MC.PRESETS)
for i in range(n_cases):
    w, h = 8 * int(rng.integers(16, 56)), 8 * int(rng.integers(12, 40))
    kind = int(rng.integers(0, 5)); name = names[int(rng.integers(len(names)))]
    nl = int(rng.integers(1, 3)); tl = int(rng.integers(0, 5))
    if fast: w, name = 64 * int(rng.integers(2, 8)), "c3_2160p_m8"
    pics = [T.PaPic(f) for f in content(kind, w, h)]
    p = MC.preset(name, nl, tl)
    if c5:
        p = MC.preset_c5(nl, tl & 3)
        p.search_area_width, p.search_area_height = [(64, 64), (64, 64), (48, 40), (56, 64), (64, 24), (40, 56), (61, 33), (24, 64)][int(rng.integers(8))]
        p.cu8x8_mode = int(rng.integers(0, 2))
        p.fractional_search_method = [2, 2, 0, 1][int(rng.integers(4))]
        name = "c5 %dx%d cu8=%d method=%d" % (p.search_area_width, p.search_area_height, p.cu8x8_mode, p.fractional_search_method)
    if nl == 2 and rng.integers(0, 4) == 0: p.same_ref_poc = 1
    r1 = pics[2] if nl == 2 else None
    o, _ = T.oracle_me_picture(pics[1], pics[0], r1, p)
    g = hip(pics[1], pics[0], r1, p)
    if fast and not os.environ.get("SVT_HIP_ME_NOFAST"): assert lib.svt_hip_me_last_instance(ctx) == 101, lib.svt_hip_me_last_instance(ctx)
    m = T.me_results_equal(o, g, nl)
    if m:
        bad += 1
        print("MISMATCH case", i, (w, h), "kind", kind, name, "nl", nl, "tl", tl, m[:3])
print("cases", n_cases, "mismatches", bad)
sys.exit(1 if bad else 0)
