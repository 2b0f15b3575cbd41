"""GPU diagnostic: bisect which parameter of a variant makes HIP ME differ from the oracle."""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, 'tests')
import me_configs as MC, svt_testlib as T
from test_gpu_me import hip_me_picture
B = T.B
lib = B.load(); ctx = C.c_void_p(); B.check(lib.svt_hip_ctx_create(C.byref(ctx), 0))
pics = [T.PaPic(f) for f in T.gen_clip_subpel(264, 200, 3, 5)]
def run(tag, p, nl):
    ref1 = pics[2] if nl == 2 else None
    o, _ = T.oracle_me_picture(pics[1], pics[0], ref1, p)
    g, _ = hip_me_picture(ctx, pics[1], pics[0], ref1, p)
    bad = T.me_results_equal(o, g, nl)
    print(tag, 'bad', bad)
    if bad:
        f = bad[0]; idx = np.argwhere(o[f] != g[f]); print(' n', len(idx), idx[:12].tolist())
        for sb, pu in idx[:3]: print('  ', sb, pu, o[sb, pu], g[sb, pu])
for nl, tl in ((1, 0), (2, 2)):
    base = MC.preset("c2_1080p_m8", nl, tl)
    for name, kv in (("full_sad", dict(fractional_search_method=1)), ("model0", dict(fractional_search_model=0)),
                     ("frac64", dict(fractional_search64x64=1)), ("cu8", dict(cu8x8_mode=0)),
                     ("cu8+model0", dict(cu8x8_mode=0, fractional_search_model=0)),
                     ("saw21", dict(search_area_width=21, search_area_height=5)),
                     ("frac64+model0", dict(fractional_search64x64=1, fractional_search_model=0))):
        p = MC.preset("c2_1080p_m8", nl, tl)
        for k, v in kv.items(): setattr(p, k, v)
        run(f"nl{nl} tl{tl} {name}", p, nl)
    run(f"nl{nl} tl{tl} ALL", MC.variant_full_sad_all_pus(nl, tl), nl)
