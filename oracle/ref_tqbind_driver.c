/*
 * ref_tqbind_driver.c -- harness for the transform call sites of the reference's encode pass (Source/Lib/Codec/EbEncDecProcess.c:3830,
 * 3890, 3940): the reference's own perform_coding_loop (:365-587; is_encode_pass = 1, do_recon = 1) for the Y, Cb and Cr transform block of
 * every block of an inter picture, with the EncDecContext / MACROBLOCKD / ModeInfo / QUANTS the encode pass has around those calls -- and
 * then the BINDING of this repository (integration/coding_loop_binding.h: append at every call site, one svt_hip_tq_batch for the list) on
 * the same source / prediction planes.  TEST INFRASTRUCTURE ONLY (rules: ref_me_driver.c; context set-up as ref_intra_driver.c).
 *
 * request : int32 magic 'SVTB', width, height, mi_stride, q_index, run_binding, device;  Y (W*H), U, V tight source planes; the same for
 *           the prediction; mi_rows * mi_stride svt_lf_mode_info (square blocks: sb_type 0 = four 4x4, 3, 6, 9; inter; 12 = 64x64 with four
 *           32x32 transform units as :3813-3825)
 * response: int32 n_blocks; REFERENCE: per block {uint8 plane, tx_size; uint16 x, y, eob}, its qcoeff then dqcoeff (n*n int16 each);
 *           recon Y, U, V;  int32 binding_rc;  BINDING (when run_binding): the same per-block records and coefficients, recon Y, U, V;
 *           then one double: the seconds spent inside the reference's perform_coding_loop calls (bench.py's cpu_baseline.reference_tq).
 *           run_binding = -1: timing run -- the binding's append is left out of the loop, only the timing double follows the reference side
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <time.h>

#define RTCD_C
#include "vpx_dsp_rtcd.h"
#include "vp9_rtcd.h"
#include "EbEncDecProcess.h"
#include "EbSequenceControlSet.h"
#include "EbUtility.h"
#include "vp9_encoder.h"
#include "vp9_quantize.h"
#include "vp9_blockd.h"
#include "vp9_scan.h"
#include "vp9_common_data.h"

#include "../include/svtvp9_hip.h"
#include "../integration/coding_loop_binding.h"

uint32_t eb_vp9_ASM_TYPES = 0;

void perform_coding_loop(EncDecContext *context_ptr, int16_t *residual_quant_coeff_buffer, const int residual_quant_coeff_stride, EbByte input_buffer,
                         uint16_t input_stride, EbByte pred_buffer, uint16_t pred_stride, int16_t *trans_coeff_buffer, int16_t *recon_coeff_buffer,
                         EbByte recon_buffer, uint16_t recon_stride, const int16_t *zbin_ptr, const int16_t *round_ptr, const int16_t *quant_ptr,
                         const int16_t *quant_shift_ptr, int16_t *dequant_ptr, uint16_t *eob, TX_SIZE tx_size, int plane, EB_BOOL is_encode_pass,
                         EB_BOOL do_recon);

static int rd(FILE *f, void *p, size_t n) { return fread(p, 1, n, f) == n ? 0 : -1; }
struct rec { uint8_t plane, tx_size; uint16_t x, y, eob; };

int main(int argc, char **argv) {
    if (argc < 3) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    int32_t h[7];
    if (rd(f, h, sizeof h) || h[0] != 0x42545653) return 3; /* 'SVTB' */
    const int W = h[1], H = h[2], mi_stride = h[3], q_index = h[4], run_binding = h[5], device = h[6];
    const int mi_rows = H / 8, sb_cols = (W + 63) / 64, sb_rows = (H + 63) / 64;
    const size_t ny = (size_t)W * H, nc = ny / 4, nall = ny + 2 * nc;
    /* a picture's three planes in one allocation, as an EbPictureBufferDesc has them: Y | Cb | Cr */
    uint8_t *srcp = (uint8_t *)malloc(nall), *predp = (uint8_t *)malloc(nall), *rec_a = (uint8_t *)calloc(nall, 1), *rec_b = (uint8_t *)calloc(nall, 1);
    if (rd(f, srcp, nall) || rd(f, predp, nall)) return 3;
    const size_t      n = (size_t)mi_rows * mi_stride;
    svt_lf_mode_info *cells = (svt_lf_mode_info *)malloc(n * sizeof *cells);
    if (rd(f, cells, n * sizeof *cells)) return 3;
    fclose(f);
    const size_t po[3] = {0, ny, ny + nc};

    setup_rtcd_internal(0);
    setup_rtcd_internal_vp9(0);
    VP9_COMP *cpi = (VP9_COMP *)calloc(1, sizeof *cpi);
    cpi->common.bit_depth = VPX_BITS_8;
    eb_vp9_init_quantizer(cpi);
    QUANTS *quants = &cpi->quants;

    EncDecContext *ctx = (EncDecContext *)calloc(1, sizeof *ctx);
    MACROBLOCKD   *xd  = (MACROBLOCKD *)calloc(1, sizeof *xd);
    ctx->e_mbd = xd;
    for (int p = 1; p < 3; p++) xd->plane[p].subsampling_x = xd->plane[p].subsampling_y = 1;
    ModeInfo  mi;
    ModeInfo *mip = &mi;
    xd->mi = &mip;

    const size_t cap = (size_t)(W / 4) * (H / 4) * 3 / 2;
    struct rec *ra = (struct rec *)calloc(cap, sizeof *ra), *rb = (struct rec *)calloc(cap, sizeof *rb);
    int16_t    *qa = (int16_t *)calloc(nall, 2), *dqa = (int16_t *)calloc(nall, 2);
    int16_t    *trans = (int16_t *)calloc(64 * 64, 2);
    size_t      cpos = 0;
    int32_t     nb = 0;
    double      seconds = 0.0;

    SvtHipTqBinding b;
    memset(&b, 0, sizeof b);
    b.list = (svt_tq_block *)calloc(cap, sizeof *b.list); b.capacity = (int32_t)cap; b.eob = (uint16_t *)calloc(cap, 2);
    b.src_base = srcp; b.pred_base = predp; b.recon_base = rec_b; b.plane_bytes = nall;
    b.qcoeff = (int16_t *)calloc(nall, 2); b.dqcoeff = (int16_t *)calloc(nall, 2); b.coeff_capacity = nall;
    svt_hip_bind_quant_tables(&b.qt[0], quants->y_zbin[q_index], quants->y_round[q_index], quants->y_quant[q_index], quants->y_quant_shift[q_index], &cpi->y_dequant[q_index][0]);
    svt_hip_bind_quant_tables(&b.qt[1], quants->uv_zbin[q_index], quants->uv_round[q_index], quants->uv_quant[q_index], quants->uv_quant_shift[q_index], &cpi->uv_dequant[q_index][0]);

    for (int sr = 0; sr < sb_rows; sr++)
        for (int sc = 0; sc < sb_cols; sc++)
            for (int z = 0; z < 64; z++) {
                int ur = 0, uc = 0;
                for (int k = 0; k < 3; k++) { uc |= ((z >> (2 * k)) & 1) << k; ur |= ((z >> (2 * k + 1)) & 1) << k; }
                const int x = sc * 64 + uc * 8, y = sr * 64 + ur * 8;
                if (x >= W || y >= H) continue;
                const svt_lf_mode_info *c = &cells[(size_t)(y >> 3) * mi_stride + (x >> 3)];
                const int sq = 8 * eb_vp9_num_8x8_blocks_wide_lookup[c->sb_type];
                if (c->sb_type > BLOCK_64X64 || eb_vp9_num_8x8_blocks_high_lookup[c->sb_type] * 8 != sq || !c->is_inter) return 5;
                if ((x % sq) || (y % sq)) continue;
                if (x + sq > W || y + sq > H) return 5;
                const int sub = c->sb_type == BLOCK_4X4, nq = sub ? 4 : 1, bsq = sub ? 4 : sq;
                for (int q4 = 0; q4 < nq; q4++) {
                    const int bx = x + (sub ? 4 * (q4 & 1) : 0), by = y + (sub ? 4 * (q4 >> 1) : 0);
                    memset(&mi, 0, sizeof mi);
                    mi.sb_type = (BLOCK_SIZE)c->sb_type; mi.mode = NEARESTMV; mi.ref_frame[0] = LAST_FRAME; mi.ref_frame[1] = NONE;
                    const TX_SIZE tx = blocksize_to_txsize[c->sb_type], tx_uv = sub ? TX_4X4 : blocksize_to_txsize[eb_vp9_ss_size_lookup[c->sb_type][1][1]];
                    mi.tx_size = tx;
                    ctx->block_origin_x = (uint16_t)bx; ctx->block_origin_y = (uint16_t)by; ctx->mi_col = bx >> 3; ctx->mi_row = by >> 3;
                    ctx->bmi_index = ((bx >> 2) & 1) + (((by >> 2) & 1) << 1); /* :3706 */
                    const int has_uv = !sub || q4 == 3, tu = 4 << tx, n_tu = bsq == 64 ? 4 : 1; /* :3813: a 64x64 block is four 32x32 units */
                    for (int p = 0; p < (has_uv ? 3 : 1); p++)
                        for (int t = 0; t < (p ? 1 : n_tu); t++) {
                            const int    ps = p ? W / 2 : W, txs = p ? tx_uv : tx, bs = 4 << txs;
                            const int    px = p ? (ROUND_UV(bx) >> 1) : bx + (t & 1) * tu, py = p ? (ROUND_UV(by) >> 1) : by + (t >> 1) * tu;
                            const size_t o = po[p] + (size_t)py * ps + px;
                            uint16_t     eob = 0;
                            struct timespec ta, tb;
                            clock_gettime(CLOCK_MONOTONIC, &ta);
                            perform_coding_loop(ctx, qa + cpos, bs, srcp + o, (uint16_t)ps, predp + o, (uint16_t)ps, trans, dqa + cpos, rec_a + o, (uint16_t)ps,
                                                p ? quants->uv_zbin[q_index] : quants->y_zbin[q_index], p ? quants->uv_round[q_index] : quants->y_round[q_index],
                                                p ? quants->uv_quant[q_index] : quants->y_quant[q_index], p ? quants->uv_quant_shift[q_index] : quants->y_quant_shift[q_index],
                                                p ? &cpi->uv_dequant[q_index][0] : &cpi->y_dequant[q_index][0], &eob, (TX_SIZE)txs, p, 1, 1);
                            clock_gettime(CLOCK_MONOTONIC, &tb);
                            seconds += (double)(tb.tv_sec - ta.tv_sec) + 1e-9 * (double)(tb.tv_nsec - ta.tv_nsec);
                            ra[nb].plane = (uint8_t)p; ra[nb].tx_size = (uint8_t)txs; ra[nb].x = (uint16_t)px; ra[nb].y = (uint16_t)py; ra[nb].eob = eob;
                            rb[nb] = ra[nb];
                            /* the same call site with the binding: append now, transform at the flush */
                            if (run_binding >= 0 && svt_hip_bind_coding_loop(&b, ctx, srcp + o, (uint16_t)ps, predp + o, (uint16_t)ps, rec_b + o, (uint16_t)ps, (TX_SIZE)txs, p, 1) != nb) return 6;
                            cpos += (size_t)bs * bs;
                            nb++;
                        }
                }
            }

    FILE *o = fopen(argv[2], "wb");
    if (!o) return 2;
    fwrite(&nb, 4, 1, o);
    fwrite(ra, sizeof *ra, (size_t)nb, o);
    fwrite(qa, 2, cpos, o); fwrite(dqa, 2, cpos, o);
    fwrite(rec_a, 1, nall, o);
    int32_t brc = -100;
    if (run_binding > 0) {
        if (svt_hip_ctx_create(&b.hip, device) != 0) { fprintf(stderr, "binding: %s\n", svt_hip_last_error()); brc = -101; }
        else {
            brc = svt_hip_bind_coding_loop_flush(&b);
            if (brc) fprintf(stderr, "binding: %s\n", svt_hip_last_error());
            svt_hip_ctx_destroy(b.hip);
        }
        for (int32_t i = 0; i < nb; i++) rb[i].eob = b.eob[i];
        fwrite(&brc, 4, 1, o);
        fwrite(rb, sizeof *rb, (size_t)nb, o);
        fwrite(b.qcoeff, 2, cpos, o); fwrite(b.dqcoeff, 2, cpos, o);
        fwrite(rec_b, 1, nall, o);
    } else fwrite(&brc, 4, 1, o);
    fwrite(&seconds, sizeof seconds, 1, o);
    fclose(o);
    return 0;
}
