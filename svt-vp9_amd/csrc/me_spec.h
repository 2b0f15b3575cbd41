/* me_spec.h -- compile-time specialisations of the ME kernel for the BASELINE parameter sets.
 *
 * Every field of svt_me_params that does not change from picture to picture inside a configuration (everything
 * except num_ref_lists, temporal_layer_index, hierarchical_levels, same_ref_poc) is known when the kernel is built:
 * an instance svt_me_sb_kernel<SPEC> overwrites those fields of its private parameter copy with the constants below,
 * the launcher picks the instance whose constants EQUAL the caller's parameters (me_spec_match) and the generic
 * instance (SPEC 0) otherwise -- so results cannot differ, only dead branches and runtime divisions disappear.
 * Values = svt_hip_me_params_preset() for the configuration (tests/test_me_presets.py checks they stay in sync). */
#ifndef SVT_ME_SPEC_H
#define SVT_ME_SPEC_H
#include "../../include/svtvp9_hip.h"

#define ME_SPEC_COUNT 5

struct me_spec_vals {
    uint8_t  hme, l0, l1, l2, cu8, cu16, rc, method, model, f64, single_quadrant, saw, sah;
    uint16_t regions_w, regions_h, tw, th, w0[2], h0[2], w1[2], h1[2], w2[2], h2[2];
};
/* SPEC 1: 2160p, tune OQ, enc-mode 8 (BASELINE C3 / C4)      SPEC 2: 1080p, tune OQ, enc-mode 8 (C2)      SPEC 3: <= 576p, tune OQ, enc-mode 9 (C1)
 * SPEC 4 / 5: 2160p, tune SQ, enc-mode 3 (C5): 64x64 search area, SSD fractional search on every PU, three HME levels with four
 * regions -- reference pictures refine the 8x8 PUs as well (cu8x8_mode 0: SPEC 4), the deepest temporal layer does not (SPEC 5) */
static constexpr me_spec_vals ME_SPECS[ME_SPEC_COUNT] = {
    {1, 1, 0, 0, 1, 0, 0, 0, 1, 0, 1, 8, 7, 2, 2, 64, 32, {32, 32}, {16, 16}, {0, 0}, {0, 0}, {0, 0}, {0, 0}},
    {1, 1, 1, 1, 1, 0, 0, 0, 1, 0, 0, 16, 9, 2, 2, 64, 48, {32, 32}, {24, 24}, {4, 4}, {4, 4}, {4, 4}, {2, 2}},
    {1, 1, 1, 1, 1, 0, 0, 0, 1, 0, 0, 16, 7, 2, 2, 32, 24, {16, 16}, {12, 12}, {4, 4}, {4, 4}, {4, 4}, {2, 2}},
    {1, 1, 1, 1, 0, 0, 0, 2, 0, 0, 0, 64, 64, 2, 2, 128, 80, {64, 64}, {40, 40}, {16, 16}, {16, 16}, {8, 8}, {8, 8}},
    {1, 1, 1, 1, 1, 0, 0, 2, 0, 0, 0, 64, 64, 2, 2, 128, 80, {64, 64}, {40, 40}, {16, 16}, {16, 16}, {8, 8}, {8, 8}},
};
/* register / LDS budget of an instance: the M8 / M9 instances are held to 96 VGPRs so that five workgroups share a CU; the C5
 * instances need 58.5 KB of LDS -- two workgroups per CU whatever the registers -- so they may use 256 VGPRs (no spills) */
constexpr int me_spec_waves_per_eu(int spec) { return spec >= 4 ? 2 : 5; }

/* the instances me_fast.h's driver serves: one HME region at the 1/16 level only, SUB_SAD refinement of the 32x32 / 16x16 PUs */
constexpr bool me_spec_fast(int spec) {
    return spec >= 1 && spec <= ME_SPEC_COUNT && ME_SPECS[spec - 1].hme && ME_SPECS[spec - 1].l0 && !ME_SPECS[spec - 1].l1 && !ME_SPECS[spec - 1].l2 &&
           ME_SPECS[spec - 1].single_quadrant && ME_SPECS[spec - 1].method == 0 && ME_SPECS[spec - 1].model == 1 && !ME_SPECS[spec - 1].f64 &&
           ME_SPECS[spec - 1].cu16 == 0 && ME_SPECS[spec - 1].cu8 == 1;
}

/* overwrite the configuration-constant fields of *p with the constants of SPEC (1..ME_SPEC_COUNT) */
template <int SPEC> __host__ __device__ inline void me_spec_apply(svt_me_params *p) {
    static_assert(SPEC >= 1 && SPEC <= ME_SPEC_COUNT, "no such specialisation");
    constexpr me_spec_vals v = ME_SPECS[SPEC - 1];
    p->enable_hme_flag = v.hme; p->enable_hme_level_0_flag = v.l0; p->enable_hme_level_1_flag = v.l1; p->enable_hme_level_2_flag = v.l2;
    p->cu8x8_mode = v.cu8; p->cu16x16_mode = v.cu16; p->rate_control_mode = v.rc;
    p->fractional_search_method = v.method; p->fractional_search_model = v.model; p->fractional_search64x64 = v.f64;
    p->single_hme_quadrant = v.single_quadrant; p->search_area_width = v.saw; p->search_area_height = v.sah;
    p->number_hme_search_region_in_width = v.regions_w; p->number_hme_search_region_in_height = v.regions_h;
    p->hme_level0_total_search_area_width = v.tw; p->hme_level0_total_search_area_height = v.th;
    p->hme_level0_search_area_in_width_array[0] = v.w0[0]; p->hme_level0_search_area_in_width_array[1] = v.w0[1];
    p->hme_level0_search_area_in_height_array[0] = v.h0[0]; p->hme_level0_search_area_in_height_array[1] = v.h0[1];
    p->hme_level1_search_area_in_width_array[0] = v.w1[0]; p->hme_level1_search_area_in_width_array[1] = v.w1[1];
    p->hme_level1_search_area_in_height_array[0] = v.h1[0]; p->hme_level1_search_area_in_height_array[1] = v.h1[1];
    p->hme_level2_search_area_in_width_array[0] = v.w2[0]; p->hme_level2_search_area_in_width_array[1] = v.w2[1];
    p->hme_level2_search_area_in_height_array[0] = v.h2[0]; p->hme_level2_search_area_in_height_array[1] = v.h2[1];
}

/* index of the specialisation whose constants equal *p, or 0 */
static inline int me_spec_match(const svt_me_params *p) {
    for (int s = 0; s < ME_SPEC_COUNT; s++) {
        const me_spec_vals &v = ME_SPECS[s];
        if (p->enable_hme_flag == v.hme && p->enable_hme_level_0_flag == v.l0 && p->enable_hme_level_1_flag == v.l1 && p->enable_hme_level_2_flag == v.l2 &&
            p->cu8x8_mode == v.cu8 && p->cu16x16_mode == v.cu16 && p->rate_control_mode == v.rc && p->fractional_search_method == v.method &&
            p->fractional_search_model == v.model && p->fractional_search64x64 == v.f64 && p->single_hme_quadrant == v.single_quadrant &&
            p->search_area_width == v.saw && p->search_area_height == v.sah && p->number_hme_search_region_in_width == v.regions_w &&
            p->number_hme_search_region_in_height == v.regions_h && p->hme_level0_total_search_area_width == v.tw &&
            p->hme_level0_total_search_area_height == v.th && p->hme_level0_search_area_in_width_array[0] == v.w0[0] &&
            p->hme_level0_search_area_in_width_array[1] == v.w0[1] && p->hme_level0_search_area_in_height_array[0] == v.h0[0] &&
            p->hme_level0_search_area_in_height_array[1] == v.h0[1] && p->hme_level1_search_area_in_width_array[0] == v.w1[0] &&
            p->hme_level1_search_area_in_width_array[1] == v.w1[1] && p->hme_level1_search_area_in_height_array[0] == v.h1[0] &&
            p->hme_level1_search_area_in_height_array[1] == v.h1[1] && p->hme_level2_search_area_in_width_array[0] == v.w2[0] &&
            p->hme_level2_search_area_in_width_array[1] == v.w2[1] && p->hme_level2_search_area_in_height_array[0] == v.h2[0] &&
            p->hme_level2_search_area_in_height_array[1] == v.h2[1])
            return s + 1;
    }
    return 0;
}
#endif
