/*
 * ref_intra_driver.c -- harness that codes one intra picture with the REFERENCE's own encode-pass functions, block by block as
 * encode_pass_sb does (Source/Lib/Codec/EbEncDecProcess.c:3680-4160):
 *     generate_intra_reference_samples (EbEncDecProcess.c:1128)  +  intra_prediction (Codec/EbIntraPrediction.c:16 ->
 *     eb_vp9_predict_intra_block, VPX/vp9_reconintra.c:410 -> the predictors of VPX/intrapred.c), for Y, Cb, Cr;
 *     perform_coding_loop (EbEncDecProcess.c:365) for Y, Cb, Cr with is_encode_pass = 1, do_recon = 1;
 *     eb_vp9_neighbor_array_unit_sample_write (Codec/EbNeighborArrays.c:107) of the block's reconstruction, as :4110-4160.
 * TEST INFRASTRUCTURE ONLY; compiled only in the build container against the reference's headers and linked with the reference's
 * own objects into oracle/_ref/ref_intra (rules: ref_me_driver.c -- no stand-ins; symbols of never-taken paths stay unresolved).
 *
 * The harness supplies what the EncDec kernel sets up around those calls: the block statistics record of every block is filled with
 * the expressions of eb_vp9_md_scan_all_blks (Codec/EbUtility.c:352-414) over the reference's own lookup tables (the table builder
 * itself, build_ep_block_stats, calls the yasm-only eb_vp9_Log2f_SSE2 and cannot run here); blocks are visited in z-order inside an
 * SB, which is the order of the reference's MD scan; the quantiser tables come from eb_vp9_init_quantizer, the RTCD tables from setup_rtcd_internal[_vp9](0) (`-asm 0`).  The three NeighborArrayUnit objects are
 * filled field by field with the sizes eb_vp9_neighbor_array_unit_ctor derives from the arguments of Codec/EbPictureControlSet.c:168-200
 * (the constructor itself allocates through the encoder handle's memory map, which does not exist here).
 *
 * request : int32 magic 'SVIN', width, height, mi_stride, q_index | mixed << 16; Y (W*H), U, V (W/2*H/2) tight source planes;
 *           mi_rows*mi_stride svt_lf_mode_info (sb_type 3 / 6 / 9, pad_[1] = luma mode, pad_[2] = chroma mode; sb_type 0: four 4x4
 *           luma blocks, their modes in the nibbles of pad_[1] (blocks 0, 1) and pad_[0] (2, 3));
 *           mixed: + Y, U, V tight planes holding the reconstruction of the picture's INTER blocks (square, 8x8 .. 64x64): an inter
 *           block is not coded here, its reconstruction goes to the neighbour arrays when its turn comes, as encode_pass_sb does after
 *           every block -- what the intra blocks of an inter picture then predict from is the reference's own bookkeeping
 * response: pred Y, U, V; recon Y, U, V (tight); qcoeff, dqcoeff (n_sb * 6144 int16 each, the product's position layout);
 *           eob map (uint16 per 4x4 unit: Y, U, V)
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>

#define RTCD_C
#include "vpx_dsp_rtcd.h"
#include "vp9_rtcd.h"
#include "EbEncDecProcess.h"
#include "EbSequenceControlSet.h"
#include "EbNeighborArrays.h"
#include "EbIntraPrediction.h"
#include "EbUtility.h"
#include "vp9_encoder.h"
#include "vp9_quantize.h"
#include "vp9_blockd.h"
#include "vp9_reconintra.h"
#include "vp9_common_data.h"

#include "../include/svtvp9_hip.h"

uint32_t eb_vp9_ASM_TYPES = 0;

void generate_intra_reference_samples(SequenceControlSet *sequence_control_set_ptr, EncDecContext *context_ptr, int plane);
void intra_prediction(EncDecContext *context_ptr, EbByte pred_buffer, uint16_t pred_stride, int plane);
void perform_coding_loop(EncDecContext *context_ptr, int16_t *residual_quant_coeff_buffer, const int residual_quant_coeff_stride, EbByte input_buffer,
                         uint16_t input_stride, EbByte pred_buffer, uint16_t pred_stride, int16_t *trans_coeff_buffer, int16_t *recon_coeff_buffer,
                         EbByte recon_buffer, uint16_t recon_stride, const int16_t *zbin_ptr, const int16_t *round_ptr, const int16_t *quant_ptr,
                         const int16_t *quant_shift_ptr, int16_t *dequant_ptr, uint16_t *eob, TX_SIZE tx_size, int plane, EB_BOOL is_encode_pass,
                         EB_BOOL do_recon);

static int rd(FILE *f, void *p, size_t n) { return fread(p, 1, n, f) == n ? 0 : -1; }

static NeighborArrayUnit *make_na(uint32_t max_w, uint32_t max_h) {
    NeighborArrayUnit *na = (NeighborArrayUnit *)calloc(1, sizeof *na);
    na->unit_size = 1; na->granularity_normal = 1; na->granularity_normal_log2 = 0; na->granularity_top_left = 1; na->granularity_top_left_log2 = 0;
    na->left_array_size = (uint16_t)max_h; na->top_array_size = (uint16_t)max_w; na->top_left_array_size = (uint16_t)(max_w + max_h);
    na->left_array = (uint8_t *)malloc(max_h); na->top_array = (uint8_t *)malloc(max_w); na->top_left_array = (uint8_t *)malloc(max_w + max_h);
    eb_vp9_neighbor_array_unit_reset(na);
    return na;
}

static uint32_t zorder4(int x4, int y4) {
    uint32_t v = 0;
    for (int b = 0; b < 4; b++) v |= (uint32_t)((x4 >> b) & 1) << (2 * b) | (uint32_t)((y4 >> b) & 1) << (2 * b + 1);
    return v;
}

int main(int argc, char **argv) {
    if (argc < 3) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    int32_t h[5];
    if (rd(f, h, sizeof h) || h[0] != 0x4E495653) return 3; /* 'SVIN' */
    const int W = h[1], H = h[2], mi_stride = h[3], q_index = h[4] & 0xffff, mixed = (h[4] >> 16) & 1;
    const int mi_rows = H / 8, mi_cols = W / 8, sb_cols = (W + 63) / 64, sb_rows = (H + 63) / 64, n_sb = sb_cols * sb_rows;
    const size_t ny = (size_t)W * H, nc = ny / 4;
    uint8_t *src[3] = {malloc(ny), malloc(nc), malloc(nc)}, *pred[3] = {calloc(ny, 1), calloc(nc, 1), calloc(nc, 1)}, *rec[3] = {calloc(ny, 1), calloc(nc, 1), calloc(nc, 1)};
    if (rd(f, src[0], ny) || rd(f, src[1], nc) || rd(f, src[2], nc)) return 3;
    const size_t      n = (size_t)mi_rows * mi_stride;
    svt_lf_mode_info *cells = (svt_lf_mode_info *)malloc(n * sizeof *cells);
    if (rd(f, cells, n * sizeof *cells)) return 3;
    if (mixed && (rd(f, rec[0], ny) || rd(f, rec[1], nc) || rd(f, rec[2], nc))) return 3;
    fclose(f);

    setup_rtcd_internal(0);
    setup_rtcd_internal_vp9(0);
    eb_vp9_init_intra_predictors();
    VP9_COMP *cpi = (VP9_COMP *)calloc(1, sizeof *cpi);
    cpi->common.bit_depth = VPX_BITS_8;
    eb_vp9_init_quantizer(cpi);
    QUANTS *quants = &cpi->quants;

    SequenceControlSet *scs = (SequenceControlSet *)calloc(1, sizeof *scs);
    scs->luma_width = (uint16_t)W; scs->luma_height = (uint16_t)H; scs->chroma_width = (uint16_t)(W / 2); scs->chroma_height = (uint16_t)(H / 2);
    EncDecContext *ctx = (EncDecContext *)calloc(1, sizeof *ctx);
    MACROBLOCKD   *xd  = (MACROBLOCKD *)calloc(1, sizeof *xd);
    ctx->e_mbd = xd;
    for (int p = 1; p < 3; p++) xd->plane[p].subsampling_x = xd->plane[p].subsampling_y = 1;
    ModeInfo  mi;
    ModeInfo *mip = &mi;
    xd->mi = &mip;
    ctx->intra_above_ref = (uint8_t *)calloc(1, 2 * 64 + 16);
    ctx->intra_left_ref  = (uint8_t *)calloc(1, 2 * 64 + 16);
    ctx->luma_recon_neighbor_array = make_na(MAX_PICTURE_WIDTH_SIZE, MAX_PICTURE_HEIGHT_SIZE);
    ctx->cb_recon_neighbor_array   = make_na(MAX_PICTURE_WIDTH_SIZE >> 1, MAX_PICTURE_HEIGHT_SIZE >> 1);
    ctx->cr_recon_neighbor_array   = make_na(MAX_PICTURE_WIDTH_SIZE >> 1, MAX_PICTURE_HEIGHT_SIZE >> 1);
    NeighborArrayUnit *na[3] = {ctx->luma_recon_neighbor_array, ctx->cb_recon_neighbor_array, ctx->cr_recon_neighbor_array};

    int16_t  *q = (int16_t *)calloc((size_t)n_sb * 6144, 2), *dq = (int16_t *)calloc((size_t)n_sb * 6144, 2);
    const size_t e1 = (size_t)(W / 4) * (H / 4), e2 = e1 + (size_t)(W / 8) * (H / 8), e3 = e2 + (size_t)(W / 8) * (H / 8);
    uint16_t *emap = (uint16_t *)calloc(e3, 2);
    const size_t eo[3] = {0, e1, e2};
    int16_t *resid = (int16_t *)calloc(64 * 64, 2), *trans = (int16_t *)calloc(64 * 64, 2), *dqs = (int16_t *)calloc(64 * 64, 2);

    for (int sr = 0; sr < sb_rows; sr++)
        for (int sc = 0; sc < sb_cols; sc++)
            for (int z = 0; z < 64; z++) {
                int ur = 0, uc = 0;
                for (int k = 0; k < 3; k++) { uc |= ((z >> (2 * k)) & 1) << k; ur |= ((z >> (2 * k + 1)) & 1) << k; }
                const int x = sc * 64 + uc * 8, y = sr * 64 + ur * 8;
                if (x >= W || y >= H) continue;
                const svt_lf_mode_info *b = &cells[(size_t)(y >> 3) * mi_stride + (x >> 3)];
                const int sq = 8 * eb_vp9_num_8x8_blocks_wide_lookup[b->sb_type];
                if (b->sb_type > BLOCK_64X64 || eb_vp9_num_8x8_blocks_high_lookup[b->sb_type] * 8 != sq) return 5;
                if ((x % sq) || (y % sq)) continue; /* not the block's first unit */
                if (x + sq > W || y + sq > H) return 5;
                if (mixed && b->is_inter) { /* coded elsewhere: only the bookkeeping of :4110-4160 */
                    for (int p = 0; p < 3; p++) {
                        const int ss = p ? 1 : 0, ps = p ? W / 2 : W, bs = sq >> ss;
                        eb_vp9_neighbor_array_unit_sample_write(na[p], rec[p], (uint32_t)ps, (uint32_t)(x >> ss), (uint32_t)(y >> ss), (uint32_t)(x >> ss),
                                                                (uint32_t)(y >> ss), (uint32_t)bs, (uint32_t)bs, NEIGHBOR_ARRAY_UNIT_FULL_MASK);
                    }
                    continue;
                }
                if (b->sb_type > BLOCK_32X32 || b->is_inter) return 5;
                /* an 8x8 unit of 4x4 blocks is four blocks of the reference's table (bsize BLOCK_4X4, origins 4 apart; bmi_index from the
                   origin, :3706); the chroma 4x4 rides with the last one (has_uv = is_last_quadrant, Codec/EbUtility.c:395) */
                const int sub = b->sb_type == BLOCK_4X4, nq = sub ? 4 : 1, bsq = sub ? 4 : sq;
                for (int q4 = 0; q4 < nq; q4++) {
                const int bx = x + (sub ? 4 * (q4 & 1) : 0), by = y + (sub ? 4 * (q4 >> 1) : 0);
                EpBlockStats stv;
                memset(&stv, 0, sizeof stv);
                stv.sq_size = bsq; stv.sq_size_uv = MAX(bsq >> 1, 4); stv.shape = PART_N;
                stv.origin_x = (uint8_t)(bx - sc * 64); stv.origin_y = (uint8_t)(by - sr * 64);
                stv.bwidth = stv.bheight = (uint8_t)bsq; stv.bwidth_uv = stv.bheight_uv = (uint8_t)MAX(4, bsq >> 1);
                stv.bsize = (BLOCK_SIZE)b->sb_type;
                stv.bsize_uv = sub ? BLOCK_4X4 : eb_vp9_ss_size_lookup[stv.bsize][1][1]; /* (the table's own entry for 4x4 is get_plane_block_size = BLOCK_INVALID,
                                                                                             whose transform size is read past blocksize_to_txsize[]: the 4x4 it means is set here) */
                stv.tx_size = blocksize_to_txsize[stv.bsize]; stv.tx_size_uv = sub ? TX_4X4 : blocksize_to_txsize[stv.bsize_uv];
                stv.has_uv = !sub || q4 == 3;
                stv.is_last_quadrant = sub && q4 == 3;
                const EpBlockStats *st = &stv;
                const int idx = 0, n_planes = st->has_uv ? 3 : 1;
                memset(&mi, 0, sizeof mi);
                mi.sb_type = st->bsize; mi.uv_mode = (PREDICTION_MODE)b->pad_[2];
                if (sub) {
                    mi.bmi[0].as_mode = (PREDICTION_MODE)(b->pad_[1] & 15); mi.bmi[1].as_mode = (PREDICTION_MODE)(b->pad_[1] >> 4);
                    mi.bmi[2].as_mode = (PREDICTION_MODE)(b->pad_[0] & 15); mi.bmi[3].as_mode = (PREDICTION_MODE)(b->pad_[0] >> 4);
                    mi.mode = mi.bmi[3].as_mode;
                } else mi.mode = (PREDICTION_MODE)b->pad_[1];
                mi.tx_size = st->tx_size; mi.ref_frame[0] = INTRA_FRAME; mi.ref_frame[1] = NONE;
                ctx->ep_block_stats_ptr = st; ctx->ep_block_index = (uint16_t)idx;
                ctx->block_origin_x = (uint16_t)bx; ctx->block_origin_y = (uint16_t)by;
                ctx->mi_col = bx >> 3; ctx->mi_row = by >> 3;
                ctx->bmi_index = ((bx >> 2) & 1) + (((by >> 2) & 1) << 1); /* :3706 */
                /* Codec/EbEncDecProcess.c:3708-3719 */
                xd->mb_to_top_edge    = -(((by >> 3) * MI_SIZE) * 8);
                xd->mb_to_bottom_edge = ((mi_rows - eb_vp9_num_8x8_blocks_high_lookup[st->bsize] - (by >> 3)) * MI_SIZE) * 8;
                xd->mb_to_left_edge   = -(((bx >> 3) * MI_SIZE) * 8);
                xd->mb_to_right_edge  = ((mi_cols - eb_vp9_num_8x8_blocks_wide_lookup[st->bsize] - (bx >> 3)) * MI_SIZE) * 8;
                /* plane positions: luma at the block's origin, chroma at ROUND_UV(origin) >> 1 (:3692-3701) */
                int px_[3], py_[3];
                px_[0] = bx; py_[0] = by; px_[1] = px_[2] = ROUND_UV(bx) >> 1; py_[1] = py_[2] = ROUND_UV(by) >> 1;
                for (int p = 0; p < n_planes; p++) {
                    const int ps = p ? W / 2 : W;
                    generate_intra_reference_samples(scs, ctx, p);
                    intra_prediction(ctx, pred[p] + (size_t)py_[p] * ps + px_[p], (uint16_t)ps, p);
                }
                for (int p = 0; p < n_planes; p++) {
                    const int ps = p ? W / 2 : W, px = px_[p], py = py_[p], bs = p ? st->sq_size_uv : st->sq_size;
                    const TX_SIZE tx = p ? st->tx_size_uv : st->tx_size;
                    const int sbw = p ? 32 : 64;
                    const size_t co = (size_t)(sr * sb_cols + sc) * 6144 + (p == 0 ? 0 : p == 1 ? 4096 : 5120) + zorder4((px % sbw) >> 2, (py % sbw) >> 2) * 16;
                    uint16_t eob = 0;
                    const size_t o = (size_t)py * ps + px;
                    perform_coding_loop(ctx, resid, bs, src[p] + o, (uint16_t)ps, pred[p] + o, (uint16_t)ps, trans, dqs, rec[p] + o, (uint16_t)ps,
                                        p ? quants->uv_zbin[q_index] : quants->y_zbin[q_index], p ? quants->uv_round[q_index] : quants->y_round[q_index],
                                        p ? quants->uv_quant[q_index] : quants->y_quant[q_index], p ? quants->uv_quant_shift[q_index] : quants->y_quant_shift[q_index],
                                        p ? &cpi->uv_dequant[q_index][0] : &cpi->y_dequant[q_index][0], &eob, tx, p, 1, 1);
                    memcpy(q + co, resid, (size_t)bs * bs * 2);
                    memcpy(dq + co, dqs, (size_t)bs * bs * 2);
                    emap[eo[p] + (size_t)(py >> 2) * (size_t)(ps >> 2) + (size_t)(px >> 2)] = eob;
                }
                for (int p = 0; p < n_planes; p++) { /* :4110-4160 */
                    const int ps = p ? W / 2 : W, bs = p ? st->bwidth_uv : st->bwidth;
                    eb_vp9_neighbor_array_unit_sample_write(na[p], rec[p], (uint32_t)ps, (uint32_t)px_[p], (uint32_t)py_[p], (uint32_t)px_[p],
                                                            (uint32_t)py_[p], (uint32_t)bs, (uint32_t)bs, NEIGHBOR_ARRAY_UNIT_FULL_MASK);
                }
                }
            }
    FILE *o = fopen(argv[2], "wb");
    if (!o) return 2;
    for (int p = 0; p < 3; p++) fwrite(pred[p], 1, p ? nc : ny, o);
    for (int p = 0; p < 3; p++) fwrite(rec[p], 1, p ? nc : ny, o);
    fwrite(q, 2, (size_t)n_sb * 6144, o);
    fwrite(dq, 2, (size_t)n_sb * 6144, o);
    fwrite(emap, 2, e3, o);
    fclose(o);
    return 0;
}
