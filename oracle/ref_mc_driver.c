/*
 * ref_mc_driver.c -- harness that runs the REFERENCE's inter_prediction() (Source/Lib/Codec/EbIntraPrediction.c:49,
 * the inter entry of prediction_fun_table, Codec/EbEncDecProcess.c:132) for every inter block and plane of a picture
 * described by a binary request file.  TEST INFRASTRUCTURE ONLY; compiled only in the build container against the
 * reference's headers and linked with the reference's own objects into oracle/_ref/ref_mc_frame (see
 * ref_me_driver.c for the rules followed: no stand-ins; symbols of never-taken paths stay unresolved).
 *
 * What the harness does is what the EncDec kernel does around the call (Codec/EbEncDecProcess.c:3690-3720,
 * 5431-5436, 5507): it fills the fields of EncDecContext / MACROBLOCKD that inter_prediction and
 * build_inter_predictors read -- block origin, block statistics (bsize, bsize_uv), mb_to_*_edge, the ModeInfo of the
 * block, the two reference picture descriptors, use_subpel_flag -- and lets the reference set up its own scale
 * factors (eb_vp9_setup_scale_factors_for_frame) and RTCD table (setup_rtcd_internal(asm)).
 *
 * asm_type 0 (`-asm 0`, the reference's C kernels) is what the tests use: with asm_type > 0 the table also names kernels
 * that exist only as yasm sources, which cannot be built here.
 *
 * request: int32 magic 'SVMC', mi_rows, mi_cols, mi_stride, use_subpel, asm_type,
 *          2 x { int32 y_stride, uv_stride, org_x, org_y, y_rows, uv_rows; Y plane (y_stride*y_rows), U, V (uv_stride*uv_rows) },
 *          mi_rows*mi_stride svt_mc_mode_info
 * response: pred Y (W*H), U, V (W/2*H/2), tight; then one double: the seconds the block loop took (clock_gettime around the
 *           inter_prediction calls only -- bench.py's cpu_baseline.reference_mc, free of this harness's file I/O).
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
#include "EbUtility.h"
#include "vp9_blockd.h"
#include "vp9_scale.h"
#include "vp9_common_data.h"

#include "../include/svtvp9_hip.h"

uint32_t eb_vp9_ASM_TYPES = 0;

void inter_prediction(struct EncDecContext *context_ptr, EbByte pred_buffer, uint16_t pred_stride, int plane);

static int rd(FILE *f, void *p, size_t n) { return fread(p, 1, n, f) == n ? 0 : -1; }

/* the reference's own size tables give the BLOCK_SIZE of a bw8 x bh8 block and its 4:2:0 chroma size */
static BLOCK_SIZE bsize_of(int bw8, int bh8) {
    for (int b = BLOCK_8X8; b < BLOCK_SIZES; b++)
        if (eb_vp9_num_8x8_blocks_wide_lookup[b] == bw8 && eb_vp9_num_8x8_blocks_high_lookup[b] == bh8) return (BLOCK_SIZE)b;
    return BLOCK_INVALID;
}

int main(int argc, char **argv) {
    if (argc < 3) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    int32_t h[6];
    if (rd(f, h, sizeof h) || h[0] != 0x434D5653) return 3; /* 'SVMC' */
    const int mi_rows = h[1], mi_cols = h[2], mi_stride = h[3], use_subpel = h[4], asm_type = h[5];
    const int W = mi_cols * 8, H = mi_rows * 8;
    eb_vp9_ASM_TYPES = (uint32_t)asm_type;
    setup_rtcd_internal((uint32_t)asm_type);

    EbPictureBufferDesc refd[2];
    memset(refd, 0, sizeof refd);
    for (int l = 0; l < 2; l++) {
        int32_t g[6];
        if (rd(f, g, sizeof g)) return 3;
        refd[l].stride_y = (uint16_t)g[0]; refd[l].stride_cb = refd[l].stride_cr = (uint16_t)g[1];
        refd[l].origin_x = (uint16_t)g[2]; refd[l].origin_y = (uint16_t)g[3];
        refd[l].width = (uint16_t)W; refd[l].height = (uint16_t)H;
        const size_t ny = (size_t)g[0] * g[4], nuv = (size_t)g[1] * g[5];
        refd[l].buffer_y = (EbByte)malloc(ny); refd[l].buffer_cb = (EbByte)malloc(nuv); refd[l].buffer_cr = (EbByte)malloc(nuv);
        if (rd(f, refd[l].buffer_y, ny) || rd(f, refd[l].buffer_cb, nuv) || rd(f, refd[l].buffer_cr, nuv)) return 3;
    }
    const size_t      n = (size_t)mi_rows * mi_stride;
    svt_mc_mode_info *cells = (svt_mc_mode_info *)malloc(n * sizeof *cells);
    if (rd(f, cells, n * sizeof *cells)) return 3;
    fclose(f);

    uint8_t *pred[3] = {(uint8_t *)calloc((size_t)W * H, 1), (uint8_t *)calloc((size_t)W * H / 4, 1), (uint8_t *)calloc((size_t)W * H / 4, 1)};

    EncDecContext *ctx = (EncDecContext *)calloc(1, sizeof *ctx);
    MACROBLOCKD   *xd  = (MACROBLOCKD *)calloc(1, sizeof *xd);
    struct scale_factors sf;
    memset(&sf, 0, sizeof sf);
    eb_vp9_setup_scale_factors_for_frame(&sf, W, H, W, H); /* as Codec/EbEncDecProcess.c:5431 */
    ctx->e_mbd = xd; ctx->sf = &sf;
    ctx->ref_pic_list[0] = &refd[0]; ctx->ref_pic_list[1] = &refd[1];
    ctx->use_subpel_flag = (uint8_t)use_subpel;
    xd->plane[0].subsampling_x = xd->plane[0].subsampling_y = 0;
    for (int p = 1; p < 3; p++) xd->plane[p].subsampling_x = xd->plane[p].subsampling_y = 1;
    ModeInfo    mi;
    ModeInfo   *mip = &mi;
    EpBlockStats st;
    xd->mi = &mip;
    ctx->ep_block_stats_ptr = &st;

    struct timespec t_begin, t_end;
    clock_gettime(CLOCK_MONOTONIC, &t_begin);
    for (int r = 0; r < mi_rows; r++)
        for (int c = 0; c < mi_cols; c++) {
            const svt_mc_mode_info *m = &cells[(size_t)r * mi_stride + c];
            if (m->ref_list[0] < 0 || m->bw8 < 1 || m->bh8 < 1 || (r % m->bh8) || (c % m->bw8)) continue;
            memset(&mi, 0, sizeof mi);
            memset(&st, 0, sizeof st);
            st.bsize    = bsize_of(m->bw8, m->bh8);
            st.bsize_uv = eb_vp9_ss_size_lookup[st.bsize][1][1];
            if (st.bsize == BLOCK_INVALID) return 5;
            mi.sb_type      = st.bsize;
            mi.ref_frame[0] = m->ref_list[0] == 0 ? LAST_FRAME : ALTREF_FRAME;
            mi.ref_frame[1] = m->ref_list[1] < 0 ? INTRA_FRAME : m->ref_list[1] == 0 ? LAST_FRAME : ALTREF_FRAME;
            for (int k = 0; k < 2; k++) { mi.mv[k].as_mv.row = m->mv_row[k]; mi.mv[k].as_mv.col = m->mv_col[k]; }
            mi.mode = NEWMV;
            ctx->block_origin_x = (uint16_t)(c * 8); ctx->block_origin_y = (uint16_t)(r * 8);
            ctx->mi_col = c; ctx->mi_row = r;
            /* Codec/EbEncDecProcess.c:3708-3719 */
            xd->mb_to_top_edge    = -((r * MI_SIZE) * 8);
            xd->mb_to_bottom_edge = ((mi_rows - eb_vp9_num_8x8_blocks_high_lookup[st.bsize] - r) * MI_SIZE) * 8;
            xd->mb_to_left_edge   = -((c * MI_SIZE) * 8);
            xd->mb_to_right_edge  = ((mi_cols - eb_vp9_num_8x8_blocks_wide_lookup[st.bsize] - c) * MI_SIZE) * 8;
            for (int p = 0; p < 3; p++) {
                const int ss = p ? 1 : 0, ps = p ? W / 2 : W;
                inter_prediction(ctx, pred[p] + (size_t)((r * 8) >> ss) * ps + ((c * 8) >> ss), (uint16_t)ps, p);
            }
        }
    clock_gettime(CLOCK_MONOTONIC, &t_end);
    const double seconds = (double)(t_end.tv_sec - t_begin.tv_sec) + 1e-9 * (double)(t_end.tv_nsec - t_begin.tv_nsec);
    FILE *o = fopen(argv[2], "wb");
    if (!o) return 2;
    fwrite(pred[0], 1, (size_t)W * H, o); fwrite(pred[1], 1, (size_t)W * H / 4, o); fwrite(pred[2], 1, (size_t)W * H / 4, o);
    fwrite(&seconds, sizeof seconds, 1, o);
    fclose(o);
    return 0;
}
