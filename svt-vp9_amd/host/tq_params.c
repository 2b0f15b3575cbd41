/*
 * tq_params.c -- host-side derivation of the quantiser tables the TQ batch takes (svt_quant_tables), the way the
 * reference derives them once per sequence: eb_vp9_init_quantizer (Source/Lib/VPX/vp9_quantize.c:206-265, sharpness 0),
 * invert_quant (:182-190), get_qzbin_factor (:192-204).  The step sizes are the caller's: eb_vp9_dc_quant / eb_vp9_ac_quant
 * of the q index and delta in use (VPX/vp9_quant_common.c).
 */
#include "../../include/svtvp9_hip.h"

/* quant / quant_shift such that ((x * quant >> 16) + x) * quant_shift >> 16 == x / d for the 16-bit magnitudes in use */
static void invert_step(int d, int16_t *quant, int16_t *shift) {
    unsigned t = (unsigned)d;
    int      l = 0;
    while (t > 1) { t >>= 1; l++; }
    const int m = 1 + (1 << (16 + l)) / d;
    *quant = (int16_t)(m - (1 << 16));
    *shift = (int16_t)(1 << (16 - l));
}

int32_t svt_hip_quant_tables_init(int32_t q_index, int32_t y_dc_step, int32_t dc_step, int32_t ac_step, svt_quant_tables *out) {
    if (!out || q_index < 0 || q_index > 255 || y_dc_step < 1 || dc_step < 1 || ac_step < 1) return SVT_HIP_ERR_BAD_PARAMETER;
    const int zbin_factor  = q_index == 0 ? 64 : (y_dc_step < 148 ? 84 : 80); /* get_qzbin_factor: from the luma DC step at delta 0 */
    const int round_factor = q_index == 0 ? 64 : 48;
    const int step[2] = {dc_step, ac_step};
    for (int i = 0; i < 2; i++) {
        invert_step(step[i], &out->quant[i], &out->quant_shift[i]);
        out->zbin[i]    = (int16_t)((zbin_factor * step[i] + 64) >> 7);
        out->round[i]   = (int16_t)((round_factor * step[i]) >> 7);
        out->dequant[i] = (int16_t)step[i];
    }
    return SVT_HIP_OK;
}
