"""GPU parity: HIP motion estimation (through the C ABI) vs the oracle, bit-exact."""
import ctypes as C

import numpy as np
import pytest

import me_configs as MC
import svt_testlib as T

B = T.B
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    lib = B.load()
    c = C.c_void_p()
    B.check(lib.svt_hip_ctx_create(C.byref(c), 0))
    yield c
    lib.svt_hip_ctx_destroy(c)


def hip_me_picture(ctx, cur, ref0, ref1, params):
    lib = B.load()
    w, h = cur.luma.shape[1], cur.luma.shape[0]
    nsb = T.n_sb(w, h)
    res = np.zeros((nsb, 85), dtype=B.ME_RESULT_DTYPE)
    rcme = np.zeros(nsb, dtype=np.uint32)
    dc, d0 = cur.desc(), ref0.desc()
    d1 = ref1.desc() if ref1 is not None else None
    B.check(lib.svt_hip_me_picture(ctx, C.byref(dc), C.byref(d0), C.byref(d1) if d1 is not None else None,
                                   C.byref(params), res.ctypes.data_as(C.c_void_p), rcme.ctypes.data_as(C.c_void_p)))
    return res, rcme


def _check(ctx, pics, p, nl):
    ref1 = pics[2] if nl == 2 else None
    o, orc = T.oracle_me_picture(pics[1], pics[0], ref1, p)
    g, grc = hip_me_picture(ctx, pics[1], pics[0], ref1, p)
    bad = T.me_results_equal(o, g, nl)
    if bad:
        f = bad[0]
        idx = np.argwhere(o[f] != g[f])
        detail = [(int(sb), int(pu), o[sb, pu].tolist()[:11], g[sb, pu].tolist()[:11]) for sb, pu in idx[:4]]
        raise AssertionError(f"{bad} mismatches={len(idx)} first={idx[:16].tolist()} detail={detail}")
    if p.rate_control_mode:
        assert np.array_equal(orc, grc)


@pytest.mark.parametrize("name", list(MC.PRESETS))
@pytest.mark.parametrize("nl,tl", [(1, 0), (2, 1), (2, 3), (1, 2)])
@pytest.mark.parametrize("clip", ["int", "subpel"])
def test_me_presets_vs_oracle(ctx, name, nl, tl, clip):
    w, h = (328, 200) if name != "c3_2160p_m8" else (384, 256)
    gen = T.gen_clip if clip == "int" else T.gen_clip_subpel
    pics = [T.PaPic(f) for f in gen(w, h, 3, 11)]
    _check(ctx, pics, MC.preset(name, nl, tl), nl)


@pytest.mark.parametrize("nl,tl", [(1, 0), (2, 2)])
def test_me_variants_vs_oracle(ctx, nl, tl):
    pics = [T.PaPic(f) for f in T.gen_clip_subpel(264, 200, 3, 5)]
    _check(ctx, pics, MC.variant_full_sad_all_pus(nl, tl), nl)
    _check(ctx, pics, MC.variant_l0_only_4quadrants(nl, tl), nl)
    if nl == 2:
        _check(ctx, pics, MC.variant_same_poc(tl), nl)


@pytest.mark.parametrize("nl,tl", [(1, 0), (2, 1), (2, 3)])
def test_me_c5_ssd_search_vs_oracle(ctx, nl, tl):
    """BASELINE config C5: SSD fractional search, 64x64 search area, 4 HME regions x 3 levels, 8x8 PUs refined."""
    pics = [T.PaPic(f) for f in T.gen_clip_subpel(328, 200, 3, 23)]
    _check(ctx, pics, MC.preset_c5(nl, tl), nl)


def test_me_wide_search_area_and_empty_area(ctx):
    """A 120 x 110 search area with all HME levels: the LDS scratch exceeds 64 KB (window offsets beyond 65535).  An empty
    search area is refused instead of spinning in the fused full-pel phase."""
    pics = [T.PaPic(f) for f in T.gen_clip_subpel(264, 200, 3, 5)]
    _check(ctx, pics, MC.variant_wide_search(2, 2), 2)
    big = [T.PaPic(f) for f in T.gen_shifted_pair(1920, 1088, 400, 240, 5)]   # HME windows of one batch beyond byte offset 65535
    p = MC.variant_wide_search(1, 0)
    g, _ = hip_me_picture(ctx, big[1], big[0], None, p)
    for b in (0, 240, 500):   # corner, middle, last rows
        o, _ = T.oracle_me_picture(big[1], big[0], None, p, b, b + 10)
        assert not T.me_results_equal(o[b:b + 10], g[b:b + 10], 1), b
    lib = B.load()
    for wh in ((0, 7), (8, 0)):
        p = MC.preset("c3_2160p_m8", 1, 0)
        p.search_area_width, p.search_area_height = wh
        res = np.zeros((T.n_sb(264, 200), 85), dtype=B.ME_RESULT_DTYPE)
        dc, d0 = pics[1].desc(), pics[0].desc()
        assert lib.svt_hip_me_picture(ctx, C.byref(dc), C.byref(d0), None, C.byref(p), res.ctypes.data_as(C.c_void_p), None) == -1


@pytest.mark.parametrize("wh", [(64, 32), (48, 48), (32, 64), (64, 64), (40, 56), (16, 127)])
def test_me_large_search_areas_full_pel_layouts(ctx, wh):
    """The fused full-pel phase has two lane layouts: areas of at least 2048 positions whose width is a multiple of 16 take the run-walking
    16x16-PU layout (csrc/me_core.h me_fullpel_fused16_dev), the others the 8x8-block one (width 40: two groups per iteration; 16 x 127: a
    tall, narrow area, runs numbered down the columns).  All of them against the oracle, on a clip with sub-pel motion and on flat pictures
    (every position ties: the first minimum in raster order has to come out of the key minima whatever the order the lanes visit positions in)."""
    pics = [T.PaPic(f) for f in T.gen_clip_subpel(264, 200, 3, 29)]
    p = MC.preset_c5(2, 2)
    p.search_area_width, p.search_area_height = wh
    _check(ctx, pics, p, 2)
    flat = [T.PaPic(np.full((136, 200), v, dtype=np.uint8)) for v in (90, 90, 91)]
    p1 = MC.preset_c5(1, 0)
    p1.search_area_width, p1.search_area_height = wh
    _check(ctx, flat, p1, 1)


def test_me_random_content(ctx):
    """Uniform random pictures: worst case for ties/early outs (there are none in the SAD paths)."""
    rng = np.random.default_rng(3)
    pics = [T.PaPic(rng.integers(0, 256, (192, 256), dtype=np.uint8)) for _ in range(3)]
    _check(ctx, pics, MC.preset("c3_2160p_m8", 2, 1), 2)
    flat = [T.PaPic(np.full((192, 256), v, dtype=np.uint8)) for v in (10, 10, 12)]
    _check(ctx, flat, MC.preset("c2_1080p_m8", 2, 1), 2)  # all-tie case: first minimum in raster order


def test_me_batch_layers_one_launch(ctx):
    """svt_hip_me_batch_layers_device: pictures of different temporal layers / list counts in ONE launch (per-picture
    parameters travel with the picture descriptors) -- every picture equals the oracle run with its own parameters; parameter
    sets that differ in a configuration field are refused."""
    import torch
    lib = B.load()
    dev = torch.device("cuda", 0)
    w, h = 384, 256
    frames = T.gen_clip_subpel(w, h, 4, 31)
    pics = [T.PaPic(f) for f in frames]
    keep = []

    def dev_desc(pa):
        d = B.PaPicture()
        for name, (a, pad) in zip(("full", "quarter", "sixteenth"), pa.planes()):
            t = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
            keep.append(t)
            pl = B.Plane()
            pl.buf, pl.stride, pl.origin_x, pl.origin_y = t.data_ptr(), t.shape[1], pad, pad
            pl.width, pl.height = t.shape[1] - 2 * pad, t.shape[0] - 2 * pad
            setattr(d, name, pl)
        return d

    d = [dev_desc(p_) for p_ in pics]
    nsb = T.n_sb(w, h)
    cases = [(1, 0, 3, (2, 0, 1)), (2, 0, 3, (2, 2, 0)), (1, 0, 2, (1, 3, 0)), (2, 1, 3, (2, 4, 0))]  # cur, ref0, ref1, (nl, tl, same_poc)
    n = len(cases)
    params = (B.MeParams * n)()
    for i, (_, _, _, (nl, tl, sp)) in enumerate(cases):
        p = MC.preset("c3_2160p_m8", nl, tl)
        p.same_ref_poc = sp
        params[i] = p
    cur = (B.PaPicture * n)(*[d[c_] for c_, _, _, _ in cases])
    r0 = (B.PaPicture * n)(*[d[a] for _, a, _, _ in cases])
    r1 = (B.PaPicture * n)(*[d[b] for _, _, b, _ in cases])
    res = [torch.zeros((nsb, 85 * 10), dtype=torch.int32, device=dev) for _ in range(n)]
    rp = (C.c_void_p * n)(*[t.data_ptr() for t in res])
    B.check(lib.svt_hip_me_batch_layers_device(ctx, n, cur, r0, r1, params, rp, None))
    B.check(lib.svt_hip_ctx_synchronize(ctx))
    for i, (c_, a, b, (nl, tl, sp)) in enumerate(cases):
        g = res[i].cpu().numpy().view(B.ME_RESULT_DTYPE).reshape(nsb, 85)
        o, _ = T.oracle_me_picture(pics[c_], pics[a], pics[b] if nl == 2 else None, params[i])
        assert not T.me_results_equal(o, g, nl), (i, nl, tl)
    bad = (B.MeParams * n)(*[params[i] for i in range(n)])
    bad[2].search_area_width = 16
    assert lib.svt_hip_me_batch_layers_device(ctx, n, cur, r0, r1, bad, rp, None) == -1  # SVT_HIP_ERR_BAD_PARAMETER
