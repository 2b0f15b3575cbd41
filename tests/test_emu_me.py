"""CPU: the HIP ME kernel's phase logic (serial host emulation of svt-vp9_amd/csrc/me_core.h) vs the oracle,
and the oracle vs the reference's own motion_estimate_sb when oracle/_ref is present."""
import numpy as np
import pytest

import me_configs as MC
import svt_testlib as T


def _cmp(fn_a, fn_b, pics, p, nl):
    ref1 = pics[2] if nl == 2 else None
    a, arc = fn_a(pics[1], pics[0], ref1, p)
    b, brc = fn_b(pics[1], pics[0], ref1, p)
    bad = T.me_results_equal(a, b, nl)
    assert not bad, bad
    if p.rate_control_mode:
        assert np.array_equal(arc, brc)


CASES = [(n, nl, tl, clip) for n in MC.PRESETS for (nl, tl) in [(1, 0), (2, 1), (2, 3)] for clip in ("int", "subpel")]


def _pics(name, clip):
    w, h = (328, 200) if name != "c3_2160p_m8" else (384, 256)
    gen = T.gen_clip if clip == "int" else T.gen_clip_subpel
    return [T.PaPic(f) for f in gen(w, h, 3, 11)]


@pytest.mark.parametrize("name,nl,tl,clip", CASES)
def test_kernel_emulation_vs_oracle(name, nl, tl, clip):
    _cmp(T.oracle_me_picture, T.emu_me_picture, _pics(name, clip), MC.preset(name, nl, tl), nl)


@pytest.mark.parametrize("nl,tl", [(1, 0), (2, 2)])
def test_kernel_emulation_variants(nl, tl):
    pics = [T.PaPic(f) for f in T.gen_clip_subpel(264, 200, 3, 5)]
    _cmp(T.oracle_me_picture, T.emu_me_picture, pics, MC.variant_full_sad_all_pus(nl, tl), nl)
    _cmp(T.oracle_me_picture, T.emu_me_picture, pics, MC.variant_l0_only_4quadrants(nl, tl), nl)
    if nl == 2:
        _cmp(T.oracle_me_picture, T.emu_me_picture, pics, MC.variant_same_poc(tl), nl)


@pytest.mark.parametrize("nl,tl", [(1, 0), (2, 2)])
def test_kernel_emulation_c5_ssd_search(nl, tl):
    """BASELINE config C5 (SSD fractional search, 64x64 search area, 4 HME regions x 3 levels)."""
    pics = [T.PaPic(f) for f in T.gen_clip_subpel(200, 136, 3, 23)]
    _cmp(T.oracle_me_picture, T.emu_me_picture, pics, MC.preset_c5(nl, tl), nl)


def test_kernel_emulation_wide_search_area():
    """scratch > 64 KB and HME windows beyond byte offset 65535 (SBs in the middle of a 1920 x 1088 picture: the whole 448 x 280 level-0 area lies inside the 1/16 plane)"""
    # the displacement (-400, -240) puts the best level-0 match into region 0, in rows a wrapped window offset would overwrite
    pics = [T.PaPic(f) for f in T.gen_shifted_pair(1920, 1088, 400, 240, 5)]
    p = MC.variant_wide_search(1, 0)
    o, _ = T.oracle_me_picture(pics[1], pics[0], None, p, 254, 258)
    e, _ = T.emu_me_picture(pics[1], pics[0], None, p, 254, 258)
    assert not T.me_results_equal(o[254:258], e[254:258], 1)
    assert (abs(o[254:258, 0]["x_mv_l0"] + 1600) <= 8).all() and (abs(o[254:258, 0]["y_mv_l0"] + 960) <= 8).all()


needs_ref = pytest.mark.skipif(not T.have_ref("ref_me_sb"), reason="oracle/_ref/ref_me_sb not built (reference absent)")


@needs_ref
@pytest.mark.parametrize("name,nl,tl,clip", CASES)
def test_oracle_vs_reference_me(name, nl, tl, clip):
    """The oracle is pinned against the REFERENCE's motion_estimate_sb compiled from /root/reference."""
    _cmp(T.ref_me_picture, T.oracle_me_picture, _pics(name, clip), MC.preset(name, nl, tl), nl)


@needs_ref
@pytest.mark.parametrize("nl,tl", [(1, 0), (2, 2)])
def test_oracle_vs_reference_me_variants(nl, tl):
    pics = [T.PaPic(f) for f in T.gen_clip_subpel(264, 200, 3, 5)]
    _cmp(T.ref_me_picture, T.oracle_me_picture, pics, MC.variant_full_sad_all_pus(nl, tl), nl)
    _cmp(T.ref_me_picture, T.oracle_me_picture, pics, MC.variant_l0_only_4quadrants(nl, tl), nl)
    if nl == 2:
        _cmp(T.ref_me_picture, T.oracle_me_picture, pics, MC.variant_same_poc(tl), nl)
    rng = np.random.default_rng(3)
    rnd = [T.PaPic(rng.integers(0, 256, (192, 256), dtype=np.uint8)) for _ in range(3)]
    _cmp(T.ref_me_picture, T.oracle_me_picture, rnd, MC.preset("c3_2160p_m8", nl, tl), nl)
    flat = [T.PaPic(np.full((192, 256), v, dtype=np.uint8)) for v in (10, 10, 12)]
    _cmp(T.ref_me_picture, T.oracle_me_picture, flat, MC.preset("c2_1080p_m8", nl, tl), nl)


def test_kernel_emulation_is_independent_of_stale_lds(monkeypatch):
    """Real LDS is inherited from whatever workgroup ran before.  Run the emulation with its LDS image kept
    across SBs/calls (warmed with other pictures) and with several fill patterns: results must not change.
    (Caught a one-row-short J plane that only showed on the GPU.)"""
    pics = [T.PaPic(f) for f in T.gen_clip_subpel(264, 200, 3, 5)]
    warm = [T.PaPic(f) for f in T.gen_clip_subpel(328, 200, 3, 11)]
    for nl, tl in ((1, 0), (2, 2)):
        p = MC.variant_full_sad_all_pus(nl, tl)
        ref1 = pics[2] if nl == 2 else None
        o, _ = T.oracle_me_picture(pics[1], pics[0], ref1, p)
        for mode in ("keep", "0", "255", "src"):
            monkeypatch.setenv("SVT_EMU_POISON", mode)
            if mode == "keep":
                for name in MC.PRESETS:
                    T.emu_me_picture(warm[1], warm[0], warm[2], MC.preset(name, 2, 1))
            e, _ = T.emu_me_picture(pics[1], pics[0], ref1, p)
            assert not T.me_results_equal(o, e, nl), mode
