/*
 * encdec_host.c -- host-side (plain C) companions of the picture-level EncDec driver (csrc/encdec.hip):
 *   - the normative VP9 constant tables the transform stage needs (inverse scan orders, quantiser steps, quantizer -> q index),
 *     generated from the reference's own objects (host/vp9_tables.inc, tools/gen_vp9_tables.py);
 *   - which of the stages behind mode decision the reference runs for a picture (do_recon, loop filter, reference padding):
 *     Codec/EbEncDecProcess.c:3653-3657, 4954-4959, 4989-4997, 5084-5089, 5127-5136, 5201, 5257-5263, 5633-5697;
 *   - host forms of the driver's device-side list / decision builders (the same inline text, csrc/encdec_core.h), for callers that
 *     want the lists on the host and for the CPU tests.
 */
#include <string.h>
#include "../../include/svtvp9_hip.h"
#include "../csrc/encdec_core.h"
#include "vp9_tables.inc"

const int16_t *svt_hip_vp9_iscan_tables(const uint32_t **offsets16, int32_t *entries) {
    if (offsets16) *offsets16 = k_vp9_iscan_off;
    if (entries) *entries = SVT_VP9_ISCAN_ENTRIES;
    return k_vp9_iscan;
}
int32_t svt_hip_vp9_qindex_from_qp(int32_t qp) { return qp < 0 || qp > 63 ? -1 : k_vp9_quantizer_to_qindex[qp]; }
int32_t svt_hip_vp9_dc_step(int32_t q_index) { return q_index < 0 || q_index > 255 ? -1 : k_vp9_dc_q[q_index]; }
int32_t svt_hip_vp9_ac_step(int32_t q_index) { return q_index < 0 || q_index > 255 ? -1 : k_vp9_ac_q[q_index]; }

/* luma (index 0) and chroma (index 1) tables of a q index with all deltas 0, the way the reference's sequence-level
 * eb_vp9_init_quantizer fills y_quant / uv_quant (VPX/vp9_quantize.c:206-265) */
int32_t svt_hip_quant_tables_for_qindex(int32_t q_index, svt_quant_tables out[2]) {
    if (!out || q_index < 0 || q_index > 255) return SVT_HIP_ERR_BAD_PARAMETER;
    const int dc = k_vp9_dc_q[q_index], ac = k_vp9_ac_q[q_index];
    int32_t   rc = svt_hip_quant_tables_init(q_index, dc, dc, ac, &out[0]);
    if (rc == SVT_HIP_OK) rc = svt_hip_quant_tables_init(q_index, dc, dc, ac, &out[1]);
    return rc;
}

int32_t svt_hip_encdec_flags_derive(const svt_encdec_flags_config *c, svt_encdec_flags *f) {
    if (!c || !f || c->enc_mode < 0 || c->enc_mode > 12 || c->tune < 0 || c->tune > 2) return SVT_HIP_ERR_BAD_PARAMETER;
    /* eb_vp9_signal_derivation_enc_dec_kernel_{sq,oq,vmaf}: limit_intra and allow_enc_dec_mismatch */
    const int limit_intra_from = c->tune == 0 ? 6 : 5; /* SQ: enc_mode <= 5 never limits; OQ / VMAF: <= 4 */
    f->limit_intra = c->enc_mode >= limit_intra_from && !c->is_used_as_reference;
    f->allow_enc_dec_mismatch = c->tune != 2 && c->enc_mode > 6 && c->temporal_layer_index > 0;
    /* encode_pass_sb (:3653-3657): an SB is reconstructed unless intra is limited and the SB holds no intra block -- here: per
       picture, for its inter SBs (an SB with an intra block is always reconstructed) */
    f->do_recon = !f->limit_intra || c->is_used_as_reference || c->recon_file;
    /* :5633-5648 */
    f->apply_loop_filter = c->loop_filter && !((f->allow_enc_dec_mismatch || !c->is_used_as_reference) && !c->recon_file);
    /* :5694-5697 */
    f->pad_reference = c->is_used_as_reference != 0;
    return SVT_HIP_OK;
}

/* ---- host forms of the device-side builders ---- */
int32_t svt_hip_md_default_picture(const svt_me_pu_result *results, int32_t pic_width, int32_t pic_height, uint32_t lambda, int32_t filter_level,
                                   svt_mc_mode_info *mc_mi, svt_lf_mode_info *lf_mi, int32_t mi_stride) {
    if (!results || !mc_mi || !lf_mi || pic_width < 8 || pic_height < 8 || (pic_width & 7) || (pic_height & 7)) return SVT_HIP_ERR_BAD_PARAMETER;
    const int mi_rows = pic_height >> 3, mi_cols = pic_width >> 3, sb_cols = (pic_width + 63) >> 6, sb_rows = (pic_height + 63) >> 6;
    if (mi_stride < mi_cols) return SVT_HIP_ERR_BAD_PARAMETER;
    for (int sr = 0; sr < sb_rows; sr++)
        for (int sc = 0; sc < sb_cols; sc++)
            for (int u = 0; u < 64; u++) {
                const int r = u >> 3, c = u & 7, ur = sr * 8 + r, uc = sc * 8 + c;
                if (ur >= mi_rows || uc >= mi_cols) continue;
                svt_md_default_unit(results + (size_t)(sr * sb_cols + sc) * 85, r, c, sr, sc, mi_rows, mi_cols, lambda, filter_level,
                                    &mc_mi[ur * mi_stride + uc], &lf_mi[ur * mi_stride + uc]);
            }
    return SVT_HIP_OK;
}

/* The transform blocks of n_pics pictures of one geometry, grouped by transform size; inside a size in the order picture, SB
 * (raster), unit of the SB (raster), and per unit luma, Cb, Cr -- the order the device-side builder produces.  size_count[4] and
 * (optionally) pos[] as the driver keeps them.  Returns the number of blocks, or a negative error (capacity too small, malformed
 * grid). */
int32_t svt_hip_tq_blocks_from_grid(int32_t n_pics, const svt_lf_mode_info *const *lf_mi, int32_t mi_stride, const svt_tq_pic_geom *geom,
                                    svt_tq_block *blocks, uint32_t *pos, int32_t capacity, int32_t size_count[4]) {
    if (n_pics < 1 || !lf_mi || !geom || !blocks || !pos || !size_count) return SVT_HIP_ERR_BAD_PARAMETER;
    const int mi_rows = geom[0].height >> 3, mi_cols = geom[0].width >> 3, sb_cols = (geom[0].width + 63) >> 6, sb_rows = (geom[0].height + 63) >> 6;
    int cnt[4] = {0, 0, 0, 0};
    for (int p = 0; p < n_pics; p++) {
        if (geom[p].width != geom[0].width || geom[p].height != geom[0].height) return SVT_HIP_ERR_BAD_PARAMETER;
        for (int ur = 0; ur < mi_rows; ur++)
            for (int uc = 0; uc < mi_cols; uc++) {
                const int o = svt_tq_unit_is_origin(lf_mi[p], mi_stride, mi_rows, mi_cols, ur, uc);
                if (o < 0) return SVT_HIP_ERR_BAD_PARAMETER;
                if (o) svt_tq_unit_counts(lf_mi[p], mi_stride, ur, uc, cnt);
            }
    }
    if (cnt[0] + cnt[1] + cnt[2] + cnt[3] > capacity) return SVT_HIP_ERR_NO_RESOURCES;
    uint32_t base[4] = {0, (uint32_t)cnt[0], (uint32_t)(cnt[0] + cnt[1]), (uint32_t)(cnt[0] + cnt[1] + cnt[2])};
    for (int p = 0; p < n_pics; p++)
        for (int sr = 0; sr < sb_rows; sr++)
            for (int sc = 0; sc < sb_cols; sc++)
                for (int u = 0; u < 64; u++) {
                    const int ur = sr * 8 + (u >> 3), uc = sc * 8 + (u & 7);
                    if (svt_tq_unit_is_origin(lf_mi[p], mi_stride, mi_rows, mi_cols, ur, uc) == 1)
                        svt_tq_unit_emit(lf_mi[p], mi_stride, ur, uc, &geom[p], k_vp9_iscan_off, base, blocks, pos);
                }
    for (int s = 0; s < 4; s++) size_count[s] = cnt[s];
    return cnt[0] + cnt[1] + cnt[2] + cnt[3];
}
