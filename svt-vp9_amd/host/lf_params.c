/*
 * lf_params.c -- host-side (plain C) loop-filter parameter derivation.
 *   svt_hip_lf_thresh_init   = update_sharpness + eb_vp9_loop_filter_init (VPX/vp9_loopfilter.c:221-262)
 *   svt_hip_lf_level_from_q  = eb_vp9_pick_filter_level, LPF_PICK_FROM_Q, 8-bit (VPX/vp9_picklpf.c:37-89)
 */
#include "../../include/svtvp9_hip.h"

void svt_hip_lf_thresh_init(svt_lf_thresh *t, int32_t sharpness_level) {
    if (!t) return;
    for (int lvl = 0; lvl <= 63; lvl++) {
        int block_inside_limit = lvl >> ((sharpness_level > 0) + (sharpness_level > 4));
        if (sharpness_level > 0 && block_inside_limit > 9 - sharpness_level) block_inside_limit = 9 - sharpness_level;
        if (block_inside_limit < 1) block_inside_limit = 1;
        t->lim[lvl]     = (uint8_t)block_inside_limit;
        t->mblim[lvl]   = (uint8_t)(2 * (lvl + 2) + block_inside_limit);
        t->hev_thr[lvl] = (uint8_t)(lvl >> 4);
    }
}

int32_t svt_hip_lf_level_from_q(int32_t ac_q, int32_t is_key_frame) {
    int guess = (ac_q * 20723 + 1015158 + (1 << 17)) >> 18; /* ROUND_POWER_OF_TWO(q * 20723 + 1015158, 18) */
    if (is_key_frame) guess -= 4;
    return guess < 0 ? 0 : guess > 63 ? 63 : guess;
}
