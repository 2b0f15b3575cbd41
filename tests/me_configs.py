"""ME parameter sets of the BASELINE configurations (via the product's own preset function) and a few
hand-built variants that exercise the remaining control-flow branches."""
import svt_testlib as T

B = T.B

# (name, pic size used for the preset lookup, enc_mode, tune)
PRESETS = {
    "c1_360p_m9": (640, 360, 9, 1),
    "c2_1080p_m8": (1920, 1080, 8, 1),
    "c3_2160p_m8": (3840, 2160, 8, 1),
}
# C5: 2160p enc-mode 3 tune 0 (SQ): SSD fractional search, 64x64 search area, HME levels 0-2 with 4 regions.  Its SSD
# path cannot be pinned against the reference build (the reference's SSD code calls the yasm-only Log2f), so it is
# checked HIP/emulation vs oracle only; everything else it exercises is pinned through the other configurations.
PRESET_C5 = ("c5_2160p_m3", (3840, 2160, 3, 0))


def preset_c5(num_lists, temporal_layer):
    w, h, mode, tune = PRESET_C5[1]
    return B.me_params_preset(w, h, mode, tune, num_lists, temporal_layer, 3)


def preset(name, num_lists, temporal_layer, hierarchical_levels=4):
    w, h, mode, tune = PRESETS[name]
    return B.me_params_preset(w, h, mode, tune, num_lists, temporal_layer, hierarchical_levels)


def variant_full_sad_all_pus(num_lists, temporal_layer):
    """1080p preset with FULL_SAD metric, 8x8 PUs refined + bi-predicted, all-PU refinement, 64x64 refinement,
    rate-control distortion on: covers the branches the M8/M9 presets leave cold."""
    p = preset("c2_1080p_m8", num_lists, temporal_layer)
    p.fractional_search_method = 1
    p.fractional_search_model = 0
    p.fractional_search64x64 = 1
    p.cu8x8_mode = 0
    p.rate_control_mode = 1
    p.search_area_width = 21   # odd width -> tail columns incl. the reference's address quirk
    p.search_area_height = 5
    return p


def variant_same_poc(temporal_layer):
    p = preset("c2_1080p_m8", 2, temporal_layer)
    p.same_ref_poc = 1
    return p


def variant_l0_only_4quadrants(num_lists, temporal_layer):
    p = preset("c2_1080p_m8", num_lists, temporal_layer)
    p.enable_hme_level_1_flag = 0
    p.enable_hme_level_2_flag = 0
    return p


def variant_wide_search(num_lists, temporal_layer):
    """1080p preset (four HME regions, three levels) with a 120 x 110 full-pel search area and C5-sized level-0 HME areas
    (64 x 40 per region, x 3.5 at layer 0): the LDS scratch is 100 KB and, on pictures at least ~1000 samples wide, the
    windows of one HME batch lie beyond byte offset 65535 (a 16-bit window offset wrapped here once)."""
    p = preset("c2_1080p_m8", num_lists, temporal_layer)
    p.search_area_width = 120
    p.search_area_height = 110
    for i in range(2):
        p.hme_level0_search_area_in_width_array[i] = 64
        p.hme_level0_search_area_in_height_array[i] = 40
    p.hme_level0_total_search_area_width = 128
    p.hme_level0_total_search_area_height = 80
    return p
