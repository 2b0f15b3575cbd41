/* oracle.h -- entry points of the CPU parity oracle (TEST INFRASTRUCTURE ONLY; see oracle_me.c). */
#ifndef SVT_ORACLE_H
#define SVT_ORACLE_H
#include <stdint.h>
#include "../include/svtvp9_hip.h"
#ifdef __cplusplus
extern "C" {
#endif
/* leaf kernels */
uint64_t oracle_spatial_full_distortion(const uint8_t *src, int ss, const uint8_t *rec, int rs, int w, int h);
uint32_t oracle_sad_nxm(const uint8_t *src, int src_stride, const uint8_t *ref, int ref_stride, int h, int w);
uint32_t oracle_avg_sad(const uint8_t *src, int src_stride, const uint8_t *r1, int s1, const uint8_t *r2, int s2,
                        int h, int w);
uint32_t oracle_avg_ssd(const uint8_t *src, int src_stride, const uint8_t *r1, int s1, const uint8_t *r2, int s2, int h, int w);
void oracle_sad_loop(const uint8_t *src, int src_stride, const uint8_t *ref, int ref_stride, int height, int width,
                     uint64_t *best_sad, int16_t *xc, int16_t *yc, int ref_stride_raw, int search_w, int search_h);
/* motion_estimate_sb over SBs [sb_begin, sb_end) of a picture (sb_end < 0 = all).
 * results is indexed [sb][85] over the WHOLE picture. */
int32_t svt_oracle_me_picture(const svt_pa_picture *cur, const svt_pa_picture *ref0, const svt_pa_picture *ref1,
                              const svt_me_params *params, svt_me_pu_result *results, uint32_t *rcme_distortion,
                              int32_t sb_begin, int32_t sb_end);
/* transform / quantisation */
void oracle_fwd_txfm(const int16_t *residual, int stride, int16_t *coeff, int tx_size, int tx_type, int partial32);
void oracle_quantize(const int16_t *coeff, int n, const svt_quant_tables *q, int16_t *qcoeff, int16_t *dqcoeff,
                     uint16_t *eob_out, const int16_t *iscan, int is32);
void oracle_inv_txfm_add(const int16_t *dqcoeff, uint8_t *dst, int stride, int tx_size, int tx_type, int eob);
int32_t svt_oracle_tq_batch(const uint8_t *src, const uint8_t *pred, uint8_t *recon, const svt_tq_block *blocks,
                            int32_t n_blocks, const svt_quant_tables *qtabs, const int16_t *iscan, int16_t *qcoeff,
                            int16_t *dqcoeff, uint16_t *eob);
#ifdef __cplusplus
}
#endif
/* L2: eb_vp9_build_mask_frame / eb_vp9_setup_mask (VPX/vp9_loopfilter.c:901-1040, 1548-1571) */
int32_t svt_oracle_lf_build_masks(const svt_lf_mode_info *mi, int32_t mi_stride, int32_t mi_rows, int32_t mi_cols,
                                  svt_lf_mask *lfm, int32_t lfm_stride);
/* T3: full_distortion_kernel32bit (C_DEFAULT/EbPictureOperators_C.c:288-311) and the batch with distortions */
void    svt_oracle_full_distortion32(const int16_t *coeff, const int16_t *recon_coeff, int32_t count, uint64_t out[2]);
int32_t svt_oracle_tq_batch_dist(const uint8_t *src, const uint8_t *pred, uint8_t *recon, const svt_tq_block *blocks,
                                 int32_t n_blocks, const svt_quant_tables *qtabs, const int16_t *iscan, int16_t *qcoeff,
                                 int16_t *dqcoeff, uint16_t *eob, uint64_t *dist);
/* M12: compute_zz_sad and eb_vp9_derive_similar_collocated_flag (Codec/EbMotionEstimationProcess.c:431-534, 747-783) */
int32_t svt_oracle_me_zz_sad(const svt_plane *cur16, const svt_plane *prev, int32_t input_resolution, uint32_t *zz, uint8_t *nmi);
void    svt_oracle_me_similar_collocated(const uint8_t *cur_mean, const uint16_t *cur_var, const uint8_t *ref_mean, const uint16_t *ref_var,
                                         int32_t n_sb, int32_t is_i_slice, int32_t is_used_as_reference, uint8_t *similar, uint8_t *similar_all);
int32_t svt_oracle_me_sb_stats(const svt_me_sb_stats_params *p, const svt_me_pu_result *results, const uint16_t *var,
                               const uint32_t *rcme, svt_me_sb_stats *out, uint32_t *hist, uint32_t *full_sb_count);
/* picture-analysis pre-ME stage: decimate_input_picture + padding (Codec/EbPictureAnalysisProcess.c:5010-5088) */
int32_t svt_oracle_pa_prepare(const uint8_t *luma, int32_t luma_stride, const svt_pa_picture *out, int32_t make_quarter);
/* pad_ref_and_set_flags (Codec/EbEncDecProcess.c:4822-4851): eb_vp9_generate_padding on Y, Cb, Cr of a reconstructed picture, in place */
int32_t svt_oracle_ref_pad(const svt_yuv_planes *pic, int32_t pad_x, int32_t pad_y);
/* compute_block_mean_compute_variance (Codec/EbPictureAnalysisProcess.c:2115-3356) */
void    svt_oracle_pa_mean8x8(const uint8_t *p, int32_t stride, uint64_t *mean, uint64_t *mean_sq);
int32_t svt_oracle_pa_mean_variance(const svt_plane *full, uint8_t *mean_out, uint16_t *var_out);
/* inter prediction of a picture from its mode-info grid (Codec/EbIntraPrediction.c:49-72 -> VPX/vp9_reconinter.c:102-252) */
int32_t svt_oracle_inter_pred_frame(const svt_mc_mode_info *mi, int32_t mi_stride, int32_t mi_rows, int32_t mi_cols,
                                    const svt_mc_host_ref ref[2], int32_t use_subpel, uint8_t *pred_y, uint8_t *pred_u,
                                    uint8_t *pred_v);
/* coefficient rate estimation (Codec/EbRateDistortionCost.c:55-172) */
int32_t svt_oracle_coeff_rate_batch(const int16_t *qcoeff, const svt_rate_block *blocks, int32_t n_blocks, const svt_rate_tables *t,
                                    const int16_t *scan_all, int32_t *bits);
/* intra path of the encode pass (oracle_intra.c): predictors (VPX/intrapred.c), reference samples (Codec/EbEncDecProcess.c:1128-1310),
 * one intra picture block by block (encode_pass_sb, :3680-4160) */
void    svt_oracle_intra_predict(int32_t mode, int32_t bs, int32_t have_left, int32_t have_top, const uint8_t *above, const uint8_t *left,
                                 uint8_t *dst, int32_t stride);
void    svt_oracle_intra_ref_samples(const uint8_t *plane, int32_t stride, int32_t x0, int32_t y0, int32_t bs, uint8_t *above_row, uint8_t *left_col);
int32_t svt_oracle_intra_picture(const uint8_t *src, uint8_t *pred, uint8_t *recon_buf, const uint32_t recon_off[3], const int32_t recon_stride[2],
                                 const svt_lf_mode_info *mi, int32_t mi_stride, int32_t width, int32_t height, const svt_quant_tables qt[2],
                                 const int16_t *iscan, const uint32_t iscan_off[16], int16_t *qcoeff, int16_t *dqcoeff, uint16_t *eob_map, int32_t mixed);
#endif
