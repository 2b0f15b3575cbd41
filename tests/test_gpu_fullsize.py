"""GPU parity at BASELINE.json's full picture sizes (3840x2160, 1920x1080 and C1's 640x360): every stage of the path through the C ABI
against the oracle on whole pictures, plus size-independent properties (specialised vs generic ME instance, block order
independence of the TQ batch, row-band independence of inter prediction)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

import me_configs as MC
import svt_testlib as T
from test_gpu_me import hip_me_picture

B = T.B
pytestmark = pytest.mark.gpu
SIZES = {"2160p": (3840, 2160, "c3_2160p_m8"), "1080p": (1920, 1080, "c2_1080p_m8"), "360p": (640, 360, "c1_360p_m9")}


@pytest.fixture(scope="module")
def ctx():
    lib = B.load()
    c = C.c_void_p()
    B.check(lib.svt_hip_ctx_create(C.byref(c), 0))
    yield c
    lib.svt_hip_ctx_destroy(c)


@pytest.fixture(scope="module")
def clips():
    return {k: [T.PaPic(f) for f in T.gen_clip_subpel(w, h, 3, 5)] for k, (w, h, _) in SIZES.items()}


@pytest.mark.parametrize("size,nl,tl", [("2160p", 2, 4), ("2160p", 2, 0), ("1080p", 2, 2), ("1080p", 1, 0), ("360p", 2, 3), ("360p", 1, 0),
                                        ("360p", 2, 1)])
def test_me_full_picture_vs_oracle(ctx, clips, size, nl, tl):
    """all 2040 (510, 60) superblocks, incl. the partial bottom row of 2160 = 33.75 x 64 (360 = 5.625 x 64); 360p = BASELINE
    config C1 at its own size with its own preset (enc-mode 9, tune 1)"""
    pics = clips[size]
    p = MC.preset(SIZES[size][2], nl, tl)
    ref1 = pics[2] if nl == 2 else None
    o, _ = T.oracle_me_picture(pics[1], pics[0], ref1, p)
    g, _ = hip_me_picture(ctx, pics[1], pics[0], ref1, p)
    assert not T.me_results_equal(o, g, nl)
    assert len(np.unique(g["x_mv_l0"])) > 8  # real motion was found


def test_me_c5_full_4k_picture_vs_oracle(ctx):
    """BASELINE config C5 at its own size: 3840x2160, enc-mode 3 tune 0 -- 64x64 full-pel search, 4 HME regions x 3 levels, SSD
    fractional search on all 85 PUs -- one whole B picture (generic kernel instance) against the oracle (host threads)."""
    pics = [T.PaPic(f) for f in T.gen_clip_subpel(3840, 2160, 3, 23)]
    p = MC.preset_c5(2, 1)
    assert (p.search_area_width, p.search_area_height, p.fractional_search_method, p.cu8x8_mode) == (64, 64, 2, 0)
    o, _ = T.oracle_me_picture_mt(pics[1], pics[0], pics[2], p)
    g, _ = hip_me_picture(ctx, pics[1], pics[0], pics[2], p)
    assert not T.me_results_equal(o, g, 2)
    assert len(np.unique(g["x_mv_l0"])) > 8


def test_me_c5_non_reference_layer_two_launches_at_4k(ctx):
    """C5's non-reference layer (cu8x8_mode 1) runs with the compact LDS layout -- two workgroups per CU -- and a second launch with the
    full layout for the SBs whose clipped search area has tail columns (csrc/me_layout.h): SBs near the right border.  One whole B
    picture against the oracle."""
    pics = [T.PaPic(f) for f in T.gen_clip_subpel(3840, 2160, 3, 31)]
    p = MC.preset_c5(2, 3)
    assert (p.search_area_width, p.search_area_height, p.fractional_search_method, p.cu8x8_mode) == (64, 64, 2, 1)
    o, _ = T.oracle_me_picture_mt(pics[1], pics[0], pics[2], p)
    g, _ = hip_me_picture(ctx, pics[1], pics[0], pics[2], p)
    assert not T.me_results_equal(o, g, 2)
    assert len(np.unique(g["x_mv_l0"])) > 8
    lib = B.load()
    lib.svt_hip_me_last_instance.argtypes = [C.c_void_p]
    assert lib.svt_hip_me_last_instance(ctx) in (204, 205)   # a specialised instance, + 200: the compact layout's pair of launches


def test_me_specialised_and_generic_instances_agree_at_4k():
    """the kernel instance specialised for the 2160p M8 parameters and the generic instance (SVT_HIP_ME_GENERIC=1) are
    the same algorithm: equal checksums of all results of a 4K B picture (run in two fresh processes: the choice is
    latched when the library first launches)"""
    code = ("import sys, zlib; sys.path.insert(0, 'tests'); import ctypes as C, numpy as np, torch; torch.cuda.init();"
            "import svt_testlib as T, me_configs as MC; from test_gpu_me import hip_me_picture; B = T.B; lib = B.load();"
            "c = C.c_void_p(); B.check(lib.svt_hip_ctx_create(C.byref(c), 0));"
            "pics = [T.PaPic(f) for f in T.gen_clip_subpel(3840, 2160, 3, 9)];"
            "g, _ = hip_me_picture(c, pics[1], pics[0], pics[2], MC.preset('c3_2160p_m8', 2, 3));"
            "print('CRC', zlib.crc32(np.ascontiguousarray(g).view(np.uint8).tobytes()))")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = []
    for generic in (False, True):
        env = dict(os.environ)
        env.pop("SVT_HIP_ME_GENERIC", None)
        if generic:
            env["SVT_HIP_ME_GENERIC"] = "1"
        r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        out.append([l for l in r.stdout.splitlines() if l.startswith("CRC")][-1])
    assert out[0] == out[1]


@pytest.mark.parametrize("size", ["2160p", "1080p"])
def test_tq_full_plane_vs_oracle_and_order_independence(ctx, size):
    w, h, _ = SIZES[size]
    case = T.make_tq_case(3, width=w, height=h - h % 32)
    o = T.oracle_tq_batch(case)
    g = T.hip_tq_batch(ctx, case)
    for n, a, b in zip(("recon", "qcoeff", "dqcoeff", "eob"), o, g):
        assert np.array_equal(a, b), (n, int(np.sum(a != b)))
    # property: a block's outputs do not depend on its place in the batch (reverse the blocks inside every size group)
    rev = dict(case)
    blocks = case["blocks"].copy()
    pos = 0
    for cnt in case["counts"]:
        blocks[pos:pos + cnt] = blocks[pos:pos + cnt][::-1]
        pos += cnt
    rev["blocks"] = blocks
    g2 = T.hip_tq_batch(ctx, rev)
    assert np.array_equal(g2[0], g[0]) and np.array_equal(g2[1], g[1]) and np.array_equal(g2[2], g[2])
    pos = 0
    for cnt in case["counts"]:
        assert np.array_equal(g2[3][pos:pos + cnt][::-1], g[3][pos:pos + cnt])
        pos += cnt


@pytest.mark.parametrize("size", ["2160p", "1080p", "360p"])
def test_lf_full_picture_vs_oracle(ctx, size):
    w, h, _ = SIZES[size]
    case = T.make_lf_case(3, w, h)
    for a, b in zip(T.hip_lf_frame(ctx, case), T.oracle_lf_frame(case)):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("size", ["2160p", "1080p", "360p"])
def test_mc_full_picture_vs_oracle_and_band_independence(ctx, size):
    w, h, _ = SIZES[size]
    case = T.make_mc_case(3, width=w, height=h, mv_range=64)
    g = T.hip_mc_frame(ctx, case)
    for a, b in zip(g, T.oracle_mc_frame(case)):
        assert np.array_equal(a, b)
    # property: prediction is local -- making the lower half of the picture intra leaves the upper half unchanged
    half = dict(case)
    mi = case["mi"].copy()
    cut = (case["mi_rows"] // 16) * 8
    mi[cut:]["ref_list"] = -1
    half["mi"] = mi
    g2 = T.hip_mc_frame(ctx, half)
    assert np.array_equal(g2[0][:cut * 8], g[0][:cut * 8]) and np.array_equal(g2[1][:cut * 4], g[1][:cut * 4])
    assert (g2[0][cut * 8:] == 0x5A).all()
