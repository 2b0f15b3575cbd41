/* me_spec.h -- compile-time specialisations of the ME kernel for the BASELINE parameter sets.
 *
 * Every field of svt_me_params that does not change from picture to picture inside a configuration (everything
 * except num_ref_lists, temporal_layer_index, hierarchical_levels, same_ref_poc) is known when the kernel is built:
 * an instance svt_me_sb_kernel<SPEC> overwrites those fields of its private parameter copy with the constants below,
 * the launcher picks the instance whose constants EQUAL the caller's parameters (me_spec_match) and the generic
 * instance (SPEC 0) otherwise -- so results cannot differ, only dead branches and runtime divisions disappear.
 * Values = svt_hip_me_params_preset() for the configuration (tests/test_abi.py checks they stay in sync). */
#ifndef SVT_ME_SPEC_H
#define SVT_ME_SPEC_H
#include "../../include/svtvp9_hip.h"

#define ME_SPEC_COUNT 3

/* overwrite the configuration-constant fields of *p with the constants of SPEC (1..ME_SPEC_COUNT) */
template <int SPEC> __host__ __device__ inline void me_spec_apply(svt_me_params *p) {
    if constexpr (SPEC == 1) { /* 2160p, tune OQ, enc-mode 8 (BASELINE C3/C4) */
        p->enable_hme_flag = 1; p->enable_hme_level_0_flag = 1; p->enable_hme_level_1_flag = 0;
        p->enable_hme_level_2_flag = 0; p->cu8x8_mode = 1; p->cu16x16_mode = 0; p->rate_control_mode = 0;
        p->fractional_search_method = 0; p->fractional_search_model = 1; p->fractional_search64x64 = 0;
        p->single_hme_quadrant = 1; p->search_area_width = 8; p->search_area_height = 7;
        p->number_hme_search_region_in_width = 2; p->number_hme_search_region_in_height = 2;
        p->hme_level0_total_search_area_width = 64; p->hme_level0_total_search_area_height = 32;
        p->hme_level0_search_area_in_width_array[0] = 32; p->hme_level0_search_area_in_width_array[1] = 32;
        p->hme_level0_search_area_in_height_array[0] = 16; p->hme_level0_search_area_in_height_array[1] = 16;
        p->hme_level1_search_area_in_width_array[0] = 0; p->hme_level1_search_area_in_width_array[1] = 0;
        p->hme_level1_search_area_in_height_array[0] = 0; p->hme_level1_search_area_in_height_array[1] = 0;
        p->hme_level2_search_area_in_width_array[0] = 0; p->hme_level2_search_area_in_width_array[1] = 0;
        p->hme_level2_search_area_in_height_array[0] = 0; p->hme_level2_search_area_in_height_array[1] = 0;
    }
    if constexpr (SPEC == 2) { /* 1080p, tune OQ, enc-mode 8 (BASELINE C2) */
        p->enable_hme_flag = 1; p->enable_hme_level_0_flag = 1; p->enable_hme_level_1_flag = 1;
        p->enable_hme_level_2_flag = 1; p->cu8x8_mode = 1; p->cu16x16_mode = 0; p->rate_control_mode = 0;
        p->fractional_search_method = 0; p->fractional_search_model = 1; p->fractional_search64x64 = 0;
        p->single_hme_quadrant = 0; p->search_area_width = 16; p->search_area_height = 9;
        p->number_hme_search_region_in_width = 2; p->number_hme_search_region_in_height = 2;
        p->hme_level0_total_search_area_width = 64; p->hme_level0_total_search_area_height = 48;
        p->hme_level0_search_area_in_width_array[0] = 32; p->hme_level0_search_area_in_width_array[1] = 32;
        p->hme_level0_search_area_in_height_array[0] = 24; p->hme_level0_search_area_in_height_array[1] = 24;
        p->hme_level1_search_area_in_width_array[0] = 4; p->hme_level1_search_area_in_width_array[1] = 4;
        p->hme_level1_search_area_in_height_array[0] = 4; p->hme_level1_search_area_in_height_array[1] = 4;
        p->hme_level2_search_area_in_width_array[0] = 4; p->hme_level2_search_area_in_width_array[1] = 4;
        p->hme_level2_search_area_in_height_array[0] = 2; p->hme_level2_search_area_in_height_array[1] = 2;
    }
    if constexpr (SPEC == 3) { /* <=576p, tune OQ, enc-mode 9 (BASELINE C1) */
        p->enable_hme_flag = 1; p->enable_hme_level_0_flag = 1; p->enable_hme_level_1_flag = 1;
        p->enable_hme_level_2_flag = 1; p->cu8x8_mode = 1; p->cu16x16_mode = 0; p->rate_control_mode = 0;
        p->fractional_search_method = 0; p->fractional_search_model = 1; p->fractional_search64x64 = 0;
        p->single_hme_quadrant = 0; p->search_area_width = 16; p->search_area_height = 7;
        p->number_hme_search_region_in_width = 2; p->number_hme_search_region_in_height = 2;
        p->hme_level0_total_search_area_width = 32; p->hme_level0_total_search_area_height = 24;
        p->hme_level0_search_area_in_width_array[0] = 16; p->hme_level0_search_area_in_width_array[1] = 16;
        p->hme_level0_search_area_in_height_array[0] = 12; p->hme_level0_search_area_in_height_array[1] = 12;
        p->hme_level1_search_area_in_width_array[0] = 4; p->hme_level1_search_area_in_width_array[1] = 4;
        p->hme_level1_search_area_in_height_array[0] = 4; p->hme_level1_search_area_in_height_array[1] = 4;
        p->hme_level2_search_area_in_width_array[0] = 4; p->hme_level2_search_area_in_width_array[1] = 4;
        p->hme_level2_search_area_in_height_array[0] = 2; p->hme_level2_search_area_in_height_array[1] = 2;
    }
}

/* index of the specialisation whose constants equal *p, or 0 */
static inline int me_spec_match(const svt_me_params *p) {
    if (p->enable_hme_flag == 1 &&
        p->enable_hme_level_0_flag == 1 &&
        p->enable_hme_level_1_flag == 0 &&
        p->enable_hme_level_2_flag == 0 &&
        p->cu8x8_mode == 1 &&
        p->cu16x16_mode == 0 &&
        p->rate_control_mode == 0 &&
        p->fractional_search_method == 0 &&
        p->fractional_search_model == 1 &&
        p->fractional_search64x64 == 0 &&
        p->single_hme_quadrant == 1 &&
        p->search_area_width == 8 &&
        p->search_area_height == 7 &&
        p->number_hme_search_region_in_width == 2 &&
        p->number_hme_search_region_in_height == 2 &&
        p->hme_level0_total_search_area_width == 64 &&
        p->hme_level0_total_search_area_height == 32 &&
        p->hme_level0_search_area_in_width_array[0] == 32 &&
        p->hme_level0_search_area_in_width_array[1] == 32 &&
        p->hme_level0_search_area_in_height_array[0] == 16 &&
        p->hme_level0_search_area_in_height_array[1] == 16 &&
        p->hme_level1_search_area_in_width_array[0] == 0 &&
        p->hme_level1_search_area_in_width_array[1] == 0 &&
        p->hme_level1_search_area_in_height_array[0] == 0 &&
        p->hme_level1_search_area_in_height_array[1] == 0 &&
        p->hme_level2_search_area_in_width_array[0] == 0 &&
        p->hme_level2_search_area_in_width_array[1] == 0 &&
        p->hme_level2_search_area_in_height_array[0] == 0 &&
        p->hme_level2_search_area_in_height_array[1] == 0)
        return 1;
    if (p->enable_hme_flag == 1 &&
        p->enable_hme_level_0_flag == 1 &&
        p->enable_hme_level_1_flag == 1 &&
        p->enable_hme_level_2_flag == 1 &&
        p->cu8x8_mode == 1 &&
        p->cu16x16_mode == 0 &&
        p->rate_control_mode == 0 &&
        p->fractional_search_method == 0 &&
        p->fractional_search_model == 1 &&
        p->fractional_search64x64 == 0 &&
        p->single_hme_quadrant == 0 &&
        p->search_area_width == 16 &&
        p->search_area_height == 9 &&
        p->number_hme_search_region_in_width == 2 &&
        p->number_hme_search_region_in_height == 2 &&
        p->hme_level0_total_search_area_width == 64 &&
        p->hme_level0_total_search_area_height == 48 &&
        p->hme_level0_search_area_in_width_array[0] == 32 &&
        p->hme_level0_search_area_in_width_array[1] == 32 &&
        p->hme_level0_search_area_in_height_array[0] == 24 &&
        p->hme_level0_search_area_in_height_array[1] == 24 &&
        p->hme_level1_search_area_in_width_array[0] == 4 &&
        p->hme_level1_search_area_in_width_array[1] == 4 &&
        p->hme_level1_search_area_in_height_array[0] == 4 &&
        p->hme_level1_search_area_in_height_array[1] == 4 &&
        p->hme_level2_search_area_in_width_array[0] == 4 &&
        p->hme_level2_search_area_in_width_array[1] == 4 &&
        p->hme_level2_search_area_in_height_array[0] == 2 &&
        p->hme_level2_search_area_in_height_array[1] == 2)
        return 2;
    if (p->enable_hme_flag == 1 &&
        p->enable_hme_level_0_flag == 1 &&
        p->enable_hme_level_1_flag == 1 &&
        p->enable_hme_level_2_flag == 1 &&
        p->cu8x8_mode == 1 &&
        p->cu16x16_mode == 0 &&
        p->rate_control_mode == 0 &&
        p->fractional_search_method == 0 &&
        p->fractional_search_model == 1 &&
        p->fractional_search64x64 == 0 &&
        p->single_hme_quadrant == 0 &&
        p->search_area_width == 16 &&
        p->search_area_height == 7 &&
        p->number_hme_search_region_in_width == 2 &&
        p->number_hme_search_region_in_height == 2 &&
        p->hme_level0_total_search_area_width == 32 &&
        p->hme_level0_total_search_area_height == 24 &&
        p->hme_level0_search_area_in_width_array[0] == 16 &&
        p->hme_level0_search_area_in_width_array[1] == 16 &&
        p->hme_level0_search_area_in_height_array[0] == 12 &&
        p->hme_level0_search_area_in_height_array[1] == 12 &&
        p->hme_level1_search_area_in_width_array[0] == 4 &&
        p->hme_level1_search_area_in_width_array[1] == 4 &&
        p->hme_level1_search_area_in_height_array[0] == 4 &&
        p->hme_level1_search_area_in_height_array[1] == 4 &&
        p->hme_level2_search_area_in_width_array[0] == 4 &&
        p->hme_level2_search_area_in_width_array[1] == 4 &&
        p->hme_level2_search_area_in_height_array[0] == 2 &&
        p->hme_level2_search_area_in_height_array[1] == 2)
        return 3;
    return 0;
}
#endif
