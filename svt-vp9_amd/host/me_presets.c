/*
 * me_presets.c -- host-side (plain C) derivation of the ME parameters for the BASELINE configurations.
 *
 * Restates, for the rows that the BASELINE configurations select, what the reference computes in
 *   eb_vp9_set_me_hme_params_oq / set_me_hme_params_sq  (Codec/EbMotionEstimationProcess.c:55-324,
 *       tables Codec/EbDefinitions.h:6495-6752),
 *   eb_vp9_signal_derivation_me_kernel_oq/_sq           (Codec/EbMotionEstimationProcess.c:541-658),
 *   picture-level flags: HME enables (Codec/EbResourceCoordinationProcess.c:343-457),
 *       use_subpel_flag (Codec/EbPictureDecisionProcess.c:682-703), cu8x8_mode / cu16x16_mode (:856-870).
 * Any other configuration is filled by the caller: every field of svt_me_params is a plain copy of the
 * reference field of the same name.
 */
#include <string.h>
#include "../../include/svtvp9_hip.h"

typedef struct preset_row {
    int res_class;  /* 0: <=576p, 3: 1080p, 4: 2160p (resolution_index, EbMotionEstimationProcess.c:60-67) */
    int enc_mode, tune;
    int l1, l2;                         /* enable_hme_level1/2 (level 0 is on in all rows) */
    int tw, th, w0, w1, h0, h1;         /* HME level 0: total w/h, per-region w (right,left), h (top,bottom) */
    int l1w, l1h, l2w, l2h;             /* HME level 1 / 2 per-region sizes (both regions equal) */
    int saw, sah;                       /* full-pel search area */
} preset_row;

/* values read off the reference tables for [resolution_index][enc_mode] */
static const preset_row rows[] = {
    /* OQ (tune 1) */
    {0, 9, 1, 1, 1, 32, 24, 16, 16, 12, 12, 4, 4, 4, 2, 16, 7},
    {3, 8, 1, 1, 1, 64, 48, 32, 32, 24, 24, 4, 4, 4, 2, 16, 9},
    {4, 8, 1, 0, 0, 64, 32, 32, 32, 16, 16, 0, 0, 0, 0, 8, 7},
    /* SQ (tune 0) */
    {4, 3, 0, 1, 1, 128, 80, 64, 64, 40, 40, 16, 16, 8, 8, 64, 64},
};

int32_t svt_hip_me_params_preset(svt_me_params *p, int32_t pic_width, int32_t pic_height, int32_t enc_mode,
                                 int32_t tune, int32_t num_ref_lists, int32_t temporal_layer_index,
                                 int32_t hierarchical_levels) {
    if (!p || num_ref_lists < 1 || num_ref_lists > 2 || temporal_layer_index < 0 || temporal_layer_index > 5 ||
        hierarchical_levels < 0 || hierarchical_levels > 5)
        return SVT_HIP_ERR_BAD_PARAMETER;
    /* input_resolution classes (Codec/EbEncHandle.c derive_input_resolution): by luma sample count */
    const long samples = (long)pic_width * pic_height;
    int        res_class = samples <= 720L * 576 ? 0 : samples <= 1280L * 720 ? 1 : samples <= 1920L * 1080 ? 3 : 4;
    const preset_row *r = 0;
    for (unsigned i = 0; i < sizeof rows / sizeof rows[0]; i++)
        if (rows[i].res_class == res_class && rows[i].enc_mode == enc_mode && rows[i].tune == tune) r = &rows[i];
    if (!r) return SVT_HIP_ERR_UNSUPPORTED;
    memset(p, 0, sizeof *p);
    p->num_ref_lists        = (uint8_t)num_ref_lists;
    p->temporal_layer_index = (uint8_t)temporal_layer_index;
    p->hierarchical_levels  = (uint8_t)hierarchical_levels;
    p->enable_hme_flag = 1;
    p->enable_hme_level_0_flag = 1;
    p->enable_hme_level_1_flag = (uint8_t)r->l1;
    p->enable_hme_level_2_flag = (uint8_t)r->l2;
    /* cu8x8_mode (EbPictureDecisionProcess.c:856-867): enc_mode >= 8 -> MODE_1; enc_mode 2..6 -> MODE_0 on
       reference pictures (the deepest temporal layer is not used as reference) */
    if (enc_mode <= 1) p->cu8x8_mode = 0;
    else if (enc_mode <= 6) p->cu8x8_mode = (uint8_t)(temporal_layer_index < hierarchical_levels ? 0 : 1);
    else if (enc_mode == 7) p->cu8x8_mode = (uint8_t)(temporal_layer_index == 0 ? 0 : 1);
    else p->cu8x8_mode = 1;
    p->cu16x16_mode = 0;
    p->same_ref_poc = 0;
    p->rate_control_mode = 0;
    /* signal derivation (EbMotionEstimationProcess.c:603-658) */
    p->single_hme_quadrant      = (uint8_t)(enc_mode > 7 && res_class >= 4);
    p->fractional_search_method = (uint8_t)(enc_mode <= 4 ? SVT_SSD_SEARCH : SVT_SUB_SAD_SEARCH);
    p->fractional_search64x64   = (uint8_t)(enc_mode <= 2);
    {
        /* use_subpel_flag (picture_level_sub_pel_settings_oq, EbPictureDecisionProcess.c:682-703) */
        int sub = 1;
        if (enc_mode == 9 && res_class >= 4) sub = temporal_layer_index == 0;
        else if (enc_mode > 9) sub = res_class >= 4 ? temporal_layer_index == 0 : temporal_layer_index < hierarchical_levels;
        p->fractional_search_model = (uint8_t)(sub ? (enc_mode <= 4 ? 0 : 1) : 2);
    }
    p->search_area_width  = (uint8_t)r->saw;
    p->search_area_height = (uint8_t)r->sah;
    p->number_hme_search_region_in_width  = 2;
    p->number_hme_search_region_in_height = 2;
    p->hme_level0_total_search_area_width  = (uint16_t)r->tw;
    p->hme_level0_total_search_area_height = (uint16_t)r->th;
    p->hme_level0_search_area_in_width_array[0]  = (uint16_t)r->w0;
    p->hme_level0_search_area_in_width_array[1]  = (uint16_t)r->w1;
    p->hme_level0_search_area_in_height_array[0] = (uint16_t)r->h0;
    p->hme_level0_search_area_in_height_array[1] = (uint16_t)r->h1;
    for (int i = 0; i < 2; i++) {
        p->hme_level1_search_area_in_width_array[i]  = (uint16_t)r->l1w;
        p->hme_level1_search_area_in_height_array[i] = (uint16_t)r->l1h;
        p->hme_level2_search_area_in_width_array[i]  = (uint16_t)r->l2w;
        p->hme_level2_search_area_in_height_array[i] = (uint16_t)r->l2h;
    }
    return SVT_HIP_OK;
}
