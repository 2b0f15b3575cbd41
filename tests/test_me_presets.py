"""Host-side ME parameter derivation (svt_hip_me_params_derive / _preset, svt_hip_input_resolution) against the REFERENCE's
own eb_vp9_derive_input_resolution + eb_vp9_signal_derivation_{pre_analysis,multi_processes,me_kernel}_{sq,oq,vmaf}: every
resolution class (incl. the class boundaries and odd aspect ratios), tune 0-2, enc_mode 0-12, four temporal layers, 60 and 30
frames/s.  The committed fixture holds the reference's output (tests/gen_golden.py me_presets); with oracle/_ref present the
reference is also run live."""
import ctypes as C
import os

import numpy as np
import pytest

import svt_testlib as T

B = T.B
GOLD = os.path.join(T.GOLDEN_DIR, "me_presets_reference.npz")


def _check(req, out):
    assert len(req) == len(out) and len(req) > 3000
    for r, want in zip(req.tolist(), out.tolist()):
        got = T.product_me_preset_row(*r)
        assert got == want, (r, got, want)


def test_me_presets_vs_golden():
    g = np.load(GOLD)
    assert np.array_equal(g["req"], T.me_preset_requests())
    _check(g["req"], g["out"])


@pytest.mark.skipif(not T.have_ref("ref_me_presets"), reason="oracle/_ref/ref_me_presets not built (reference absent)")
def test_me_presets_vs_reference_live():
    r = T.ref_me_presets()
    _check(r["req"], r["out"])


def test_me_preset_shorthand_is_the_default_structure():
    """svt_hip_me_params_preset = derive() with is_used_as_reference = (layer < levels) at 60 frames/s; BASELINE rows."""
    for (w, h, mode, tune, nl, tl, hl) in ((3840, 2160, 8, 1, 2, 1, 4), (1920, 1080, 8, 1, 2, 4, 4), (640, 360, 9, 1, 1, 0, 4),
                                           (3840, 2160, 3, 0, 2, 3, 3), (2048, 1080, 8, 1, 2, 2, 4), (960, 540, 5, 2, 2, 4, 4)):
        a = B.me_params_preset(w, h, mode, tune, nl, tl, hl)
        b = B.me_params_derive(pic_width=w, pic_height=h, enc_mode=mode, tune=tune, frame_rate=60, num_ref_lists=nl,
                               temporal_layer_index=tl, hierarchical_levels=hl, is_used_as_reference=int(tl < hl))
        assert bytes(a) == bytes(b)
    p = B.me_params_preset(3840, 2160, 8, 1, 2, 1, 4)
    assert (p.search_area_width, p.search_area_height, p.single_hme_quadrant, p.enable_hme_level_1_flag) == (8, 7, 1, 0)


def test_me_preset_rejects_bad_arguments():
    q = B.MeParams()
    lib = B.load()
    assert lib.svt_hip_me_params_preset(C.byref(q), 1280, 720, 13, 1, 1, 0, 4) == -1
    assert lib.svt_hip_me_params_preset(C.byref(q), 1280, 720, 5, 3, 1, 0, 4) == -1
    assert lib.svt_hip_me_params_preset(C.byref(q), 1280, 720, 5, 1, 3, 0, 4) == -1
    assert lib.svt_hip_me_params_preset(None, 1280, 720, 5, 1, 1, 0, 4) == -1


def test_every_baseline_configuration_has_a_specialised_kernel_instance():
    """the constants compiled into svt_me_sb_kernel<SPEC> (csrc/me_spec.h) equal what the parameter derivation yields for the BASELINE
    configurations: the launcher finds an instance for each of them, every layer; any other parameter set gets the generic instance"""
    import ctypes as C
    lib = B.load()
    want = {("c1", 640, 360, 9, 1, 4): {3}, ("c2", 1920, 1080, 8, 1, 4): {2}, ("c3", 3840, 2160, 8, 1, 4): {1}, ("c5", 3840, 2160, 3, 0, 3): {4, 5}}
    for (name, w, h, mode, tune, levels), inst in want.items():
        got = set()
        for layer in range(levels + 1):
            for nl in (1, 2):
                p = B.me_params_preset(w, h, mode, tune, nl, layer, levels)
                got.add(lib.svt_hip_me_kernel_instance(C.byref(p)))
        assert got == inst, (name, got)
    p = B.me_params_preset(1280, 720, 6, 1, 2, 1, 4)
    assert lib.svt_hip_me_kernel_instance(C.byref(p)) == 0
    a, b = B.me_params_preset(3840, 2160, 3, 0, 2, 1, 3), B.me_params_preset(3840, 2160, 3, 0, 2, 3, 3)
    assert lib.svt_hip_me_params_same_launch(C.byref(a), C.byref(a)) == 1 and lib.svt_hip_me_params_same_launch(C.byref(a), C.byref(b)) == 0   # cu8x8_mode differs


def test_me_lds_budgets_of_the_baseline_configurations():
    """occupancy is LDS-bound for every ME instance: the layouts must stay inside the budgets the design counts on --
    2160p enc-mode 8: five workgroups per CU (25 granules of 1 280 bytes); 2160p enc-mode 3 (BASELINE C5), both layer kinds: two per CU with the
    compact layout (81 920 bytes each), one without; 1080p / 360p enc-mode 8 / 9: four."""
    import me_configs as MC
    lib = B.load()
    lib.svt_hip_me_lds_bytes.restype = C.c_int32
    granules = lambda b: (b + 1279) // 1280
    p = MC.preset("c3_2160p_m8", 2, 4)
    assert 128 // granules(lib.svt_hip_me_lds_bytes(C.byref(p), 0)) == 5
    for tl in (0, 3):   # reference layers (8x8 PUs refined) / the non-reference layer
        p = MC.preset_c5(2, tl)
        full, compact = lib.svt_hip_me_lds_bytes(C.byref(p), 0), lib.svt_hip_me_lds_bytes(C.byref(p), 1)
        assert 128 // granules(full) == 1 and 0 < compact <= 81920 and 128 // granules(compact) == 2, (tl, full, compact)
    for name in ("c2_1080p_m8", "c1_360p_m9"):
        p = MC.preset(name, 2, 2)
        assert 128 // granules(lib.svt_hip_me_lds_bytes(C.byref(p), 0)) == 4, name
    p = MC.preset("c3_2160p_m8", 2, 4)
    p.search_area_width = 21
    assert lib.svt_hip_me_lds_bytes(C.byref(p), 1) < 0   # no compact layout for widths that are not multiples of 8
