/*
 * me_presets.c -- host-side (plain C) derivation of the ME parameters of a picture, for every combination the reference
 * accepts: tune 0 (SQ) / 1 (OQ) / 2 (VMAF), enc_mode 0..12, any picture size.
 *
 * Mirrors what the reference computes, per picture, in
 *   eb_vp9_derive_input_resolution                        (Codec/EbSequenceControlSet.c:489-499),
 *   eb_vp9_signal_derivation_pre_analysis_{sq,oq,vmaf}    (Codec/EbResourceCoordinationProcess.c:291-460): HME enables,
 *   eb_vp9_signal_derivation_multi_processes_{sq,oq,vmaf} (Codec/EbPictureDecisionProcess.c:682-925): use_subpel_flag,
 *       cu8x8_mode, cu16x16_mode,
 *   eb_vp9_set_me_hme_params_{sq,oq,vmaf} and eb_vp9_signal_derivation_me_kernel_{sq,oq,vmaf}
 *       (Codec/EbMotionEstimationProcess.c:55-324, 541-720; tables Codec/EbDefinitions.h:6367-6752).
 * The search-area numbers (me_preset_table.inc) are produced by running those functions (tools/gen_me_preset_table.py);
 * the function is pinned against the reference for every (size class, tune, mode, layer) in tests/test_me_presets.py.
 */
#include <string.h>
#include "../../include/svtvp9_hip.h"

enum { C_L0, C_L1, C_L2, C_TW, C_TH, C_W0, C_W1, C_H0, C_H1, C_L1W0, C_L1W1, C_L1H0, C_L1H1, C_L2W0, C_L2W1, C_L2H0, C_L2H1,
       C_SAW, C_SAH, C_COUNT };

static const uint8_t preset_table[3][5][13][C_COUNT] = {
#include "me_preset_table.inc"
};

/* INPUT_SIZE_*_RANGE from the luma sample count (thresholds INPUT_SIZE_1080i/1080p/4K_TH, Codec/EbDefinitions.h:322-325) */
int32_t svt_hip_input_resolution(int32_t pic_width, int32_t pic_height) {
    const uint32_t n = (uint32_t)pic_width * (uint32_t)pic_height;
    return n < 0xB71B0u ? 0 : n < 0x1AB3F0u ? 1 : n < 0x29F630u ? 2 : 3;
}

static uint16_t max16(uint16_t a, int b) { return a > b ? a : (uint16_t)b; }

int32_t svt_hip_me_params_derive(svt_me_params *p, const svt_me_picture_config *c) {
    if (!p || !c || c->num_ref_lists < 1 || c->num_ref_lists > 2 || c->temporal_layer_index < 0 || c->temporal_layer_index > 5 ||
        c->hierarchical_levels < 0 || c->hierarchical_levels > 5 || c->tune < 0 || c->tune > 2 || c->enc_mode < 0 ||
        c->enc_mode > 12 || c->pic_width < 1 || c->pic_height < 1)
        return SVT_HIP_ERR_BAD_PARAMETER;
    const int mode = c->enc_mode, tune = c->tune, tl = c->temporal_layer_index, used = c->is_used_as_reference != 0;
    const int res   = svt_hip_input_resolution(c->pic_width, c->pic_height);
    const int ratio = c->pic_width / c->pic_height;
    /* resolution_index (EbMotionEstimationProcess.c:60-67); the VMAF tune always takes row 3 (:276) */
    int idx = res == 0 ? 0 : (res <= 1 && ratio < 2) ? 1 : (res <= 1 && ratio > 3) ? 2 : res <= 2 ? 3 : 4;
    if (tune == 2) idx = 3;
    const uint8_t *r = preset_table[tune][idx][mode];

    memset(p, 0, sizeof *p);
    p->num_ref_lists           = (uint8_t)c->num_ref_lists;
    p->temporal_layer_index    = (uint8_t)tl;
    p->hierarchical_levels     = (uint8_t)c->hierarchical_levels;
    p->enable_hme_flag         = 1; /* use_default_me_hme */
    p->enable_hme_level_0_flag = r[C_L0];
    p->enable_hme_level_1_flag = r[C_L1];
    p->enable_hme_level_2_flag = r[C_L2];
    /* cu8x8_mode, identical in the three tunes (EbPictureDecisionProcess.c:806-816, 857-867, 906-916) */
    p->cu8x8_mode   = (uint8_t)(mode <= 1 ? 0 : mode <= 6 ? !used : mode == 7 ? tl != 0 : 1);
    p->cu16x16_mode = 0;
    p->same_ref_poc = (uint8_t)(c->same_ref_poc != 0);
    p->rate_control_mode = (uint8_t)c->rate_control_mode;
    /* use_subpel_flag (picture_level_sub_pel_settings_{sq,oq,vmaf}, EbPictureDecisionProcess.c:682-751) */
    int sub;
    if (tune == 1)
        sub = mode <= 8 ? 1 : res >= 3 ? tl == 0 : mode <= 9 ? 1 : used;
    else if (tune == 2)
        sub = mode <= 8 ? 1 : res >= 3 ? tl == 0 : 1;
    else if (res >= 3)
        sub = mode <= 4 ? 1 : mode <= 7 ? used : mode <= 10 ? tl == 0 : 0;
    else
        sub = mode <= 4 ? 1 : mode <= 8 ? used : mode <= 9 ? tl == 0 : 0;
    /* ME kernel signals (EbMotionEstimationProcess.c:561-598, identical in the three tunes) */
    p->single_hme_quadrant      = (uint8_t)(mode > 7 && res >= 3);
    p->fractional_search_method = (uint8_t)(mode <= 4 ? SVT_SSD_SEARCH : SVT_SUB_SAD_SEARCH);
    p->fractional_search64x64   = (uint8_t)(mode <= 2);
    p->fractional_search_model  = (uint8_t)(sub ? (mode <= 4 ? 0 : 1) : 2);
    p->search_area_width  = r[C_SAW];
    p->search_area_height = r[C_SAH];
    p->number_hme_search_region_in_width  = 2;
    p->number_hme_search_region_in_height = 2;
    p->hme_level0_total_search_area_width  = r[C_TW];
    p->hme_level0_total_search_area_height = r[C_TH];
    p->hme_level0_search_area_in_width_array[0]  = r[C_W0];
    p->hme_level0_search_area_in_width_array[1]  = r[C_W1];
    p->hme_level0_search_area_in_height_array[0] = r[C_H0];
    p->hme_level0_search_area_in_height_array[1] = r[C_H1];
    p->hme_level1_search_area_in_width_array[0]  = r[C_L1W0];
    p->hme_level1_search_area_in_width_array[1]  = r[C_L1W1];
    p->hme_level1_search_area_in_height_array[0] = r[C_L1H0];
    p->hme_level1_search_area_in_height_array[1] = r[C_L1H1];
    p->hme_level2_search_area_in_width_array[0]  = r[C_L2W0];
    p->hme_level2_search_area_in_width_array[1]  = r[C_L2W1];
    p->hme_level2_search_area_in_height_array[0] = r[C_L2H0];
    p->hme_level2_search_area_in_height_array[1] = r[C_L2H1];
    /* HME level 0 is widened for 4K content at <= 30 frames/s (SQ :111-155, OQ :206-235; not in the VMAF tune) */
    if (res == 3 && c->frame_rate <= 30 && tune != 2) {
        int tw = 0, th = 0, w = 0, h = 0;
        if (tune == 0) {
            if (mode == 5 || mode == 6) tw = 128, th = 64, w = 64, h = 32;
            else if (mode == 7 || mode == 8) tw = 96, th = 64, w = 48, h = 32;
            else if (mode >= 9) tw = 64, th = 48, w = 32, h = 24;
        } else {
            if (mode == 6 || mode == 7) tw = 96, th = 64, w = 48, h = 32;
            else if (mode >= 8) tw = 64, th = 48, w = 32, h = 24;
        }
        p->hme_level0_total_search_area_width  = max16(p->hme_level0_total_search_area_width, tw);
        p->hme_level0_total_search_area_height = max16(p->hme_level0_total_search_area_height, th);
        for (int i = 0; i < 2; i++) {
            p->hme_level0_search_area_in_width_array[i]  = max16(p->hme_level0_search_area_in_width_array[i], w);
            p->hme_level0_search_area_in_height_array[i] = max16(p->hme_level0_search_area_in_height_array[i], h);
        }
    }
    return SVT_HIP_OK;
}

int32_t svt_hip_me_params_preset(svt_me_params *p, int32_t pic_width, int32_t pic_height, int32_t enc_mode,
                                 int32_t tune, int32_t num_ref_lists, int32_t temporal_layer_index,
                                 int32_t hierarchical_levels) {
    svt_me_picture_config c;
    memset(&c, 0, sizeof c);
    c.pic_width = pic_width; c.pic_height = pic_height; c.enc_mode = enc_mode; c.tune = tune;
    c.num_ref_lists = num_ref_lists; c.temporal_layer_index = temporal_layer_index;
    c.hierarchical_levels = hierarchical_levels;
    /* in the reference's random-access structure every layer but the deepest is referenced */
    c.is_used_as_reference = temporal_layer_index < hierarchical_levels;
    c.frame_rate = 60;
    return svt_hip_me_params_derive(p, &c);
}

/* may two parameter sets share one launch (svt_hip_me_batch_layers_device)?  They may differ in the four per-picture fields; compared
 * field by field -- the records carry padding bytes that nothing initialises */
int32_t svt_hip_me_params_same_launch(const svt_me_params *a, const svt_me_params *b) {
    if (!a || !b) return 0;
#define EQ(f) (a->f == b->f)
#define EQ2(f) (a->f[0] == b->f[0] && a->f[1] == b->f[1])
    return EQ(enable_hme_flag) && EQ(enable_hme_level_0_flag) && EQ(enable_hme_level_1_flag) && EQ(enable_hme_level_2_flag) && EQ(cu8x8_mode) &&
           EQ(cu16x16_mode) && EQ(rate_control_mode) && EQ(fractional_search_method) && EQ(fractional_search_model) && EQ(fractional_search64x64) &&
           EQ(single_hme_quadrant) && EQ(search_area_width) && EQ(search_area_height) && EQ(number_hme_search_region_in_width) &&
           EQ(number_hme_search_region_in_height) && EQ(hme_level0_total_search_area_width) && EQ(hme_level0_total_search_area_height) &&
           EQ2(hme_level0_search_area_in_width_array) && EQ2(hme_level0_search_area_in_height_array) && EQ2(hme_level1_search_area_in_width_array) &&
           EQ2(hme_level1_search_area_in_height_array) && EQ2(hme_level2_search_area_in_width_array) && EQ2(hme_level2_search_area_in_height_array);
#undef EQ
#undef EQ2
}
