/*
 * ref_lf_driver.c -- harness that runs the REFERENCE's eb_vp9_loop_filter_frame
 * (/root/reference/Source/Lib/VPX/vp9_loopfilter.c:1521) on a frame described by a binary request file.
 * TEST INFRASTRUCTURE ONLY; compiled only in the build container against the reference's headers, linked with
 * the reference's own objects into oracle/_ref/ref_lf_frame (see ref_me_driver.c for the rules followed).
 *
 * This translation unit plays the role Codec/EbEncHandle.c plays in the reference for the RTCD dispatch:
 * it includes VPX/vpx_dsp_rtcd.h / VPX/vp9_rtcd.h with RTCD_C (which defines the function pointers and the
 * reference's own setup_rtcd_internal()) and calls setup_rtcd_internal(0) = `-asm 0`, so every eb_vpx_lpf_*
 * pointer is bound to the reference's C kernel by the reference's own code.
 *
 * request: int32 magic 'SVLF', int32 width, height, y_stride, uv_stride, mi_rows, mi_cols, lfm_stride, n_lfm,
 *          y_only | sharpness_level << 8,
 *          svt_lf_thresh, n_lfm * svt_lf_mask (= LOOP_FILTER_MASK), Y plane (y_stride*height), U, V (uv_stride*height/2)
 * response: the three filtered planes, same layout.
 *
 * Second request kind (mask construction, eb_vp9_setup_mask :901 for every SB as eb_vp9_build_mask_frame :1548 does):
 * request: int32 magic 'SVLM', mi_rows, mi_cols, mi_stride, lfm_stride, then lf_info.lvl[8][4][2] bytes, then
 *          mi_rows*mi_stride cells of 6 bytes {sb_type, tx_size, skip, ref_frame[0], mode, segment_id}
 * response: (sb_rows * lfm_stride) LOOP_FILTER_MASKs.
 *
 * Third request kind (parameter derivation; rows L0 / L1 of the scope table):
 * request: int32 magic 'SVLP' (+ 4 ignored int32)
 * response: for sharpness 0..7: eb_vp9_loop_filter_init's lfthr[0..63] as {mblim[64], lim[64], hev_thr[64]} bytes (byte 0
 *           of each SIMD_WIDTH vector; the driver checks that all SIMD_WIDTH bytes are equal), then for key_frame 0,1 and
 *           base_qindex 0..255: int32 eb_vp9_ac_quant(qindex, 0, 8 bit), int32 lf.filter_level chosen by
 *           eb_vp9_pick_filter_level(cpi, LPF_PICK_FROM_Q), int32 lf.sharpness_level it left behind.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>

#define RTCD_C
#include "vpx_dsp_rtcd.h"
#include "vp9_rtcd.h"
#include "vp9_onyxc_int.h"
#include "vp9_blockd.h"
#include "vp9_loopfilter.h"
#include "vp9_quant_common.h"
#include "vp9_encoder.h"
#include "vp9_picklpf.h"

#include "../include/svtvp9_hip.h"

uint32_t eb_vp9_ASM_TYPES = 0;

static int rd(FILE *f, void *p, size_t n) { return fread(p, 1, n, f) == n ? 0 : -1; }

int main(int argc, char **argv) {
    if (argc < 3) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    int32_t h[10];
    if (rd(f, h, 5 * sizeof(int32_t))) return 3;
    if (h[0] == 0x4D4C5653) { /* 'SVLM' */
        const int mi_rows = h[1], mi_cols = h[2], mi_stride = h[3], lfm_stride = h[4];
        const int sb_rows = (mi_rows + 7) / 8;
        VP9_COMMON *cm = (VP9_COMMON *)calloc(1, sizeof *cm);
        cm->mi_rows = mi_rows; cm->mi_cols = mi_cols; cm->mi_stride = mi_stride;
        if (rd(f, cm->lf_info.lvl, sizeof cm->lf_info.lvl)) return 3;
        if (sizeof cm->lf_info.lvl != 8 * 4 * 2) { fprintf(stderr, "lf_info.lvl layout\n"); return 4; }
        const size_t n = (size_t)mi_rows * mi_stride;
        uint8_t     *cells = (uint8_t *)malloc(n * 6);
        if (rd(f, cells, n * 6)) return 3;
        fclose(f);
        ModeInfo  *mis  = (ModeInfo *)calloc(n, sizeof *mis);
        ModeInfo **grid = (ModeInfo **)calloc(n, sizeof *grid);
        for (size_t i = 0; i < n; i++) {
            mis[i].sb_type = (BLOCK_SIZE)cells[6 * i]; mis[i].tx_size = (TX_SIZE)cells[6 * i + 1]; mis[i].skip = cells[6 * i + 2];
            mis[i].ref_frame[0] = (MV_REFERENCE_FRAME)cells[6 * i + 3]; mis[i].mode = (PREDICTION_MODE)cells[6 * i + 4];
            mis[i].segment_id = cells[6 * i + 5];
            grid[i] = &mis[i];
        }
        LOOP_FILTER_MASK *lfm = (LOOP_FILTER_MASK *)calloc((size_t)sb_rows * lfm_stride, sizeof *lfm);
        cm->lf.lfm = lfm; cm->lf.lfm_stride = lfm_stride;
        for (int mi_row = 0; mi_row < mi_rows; mi_row += 8)
            for (int mi_col = 0; mi_col < mi_cols; mi_col += 8)
                eb_vp9_setup_mask(cm, mi_row, mi_col, grid + (size_t)mi_row * mi_stride + mi_col, mi_stride,
                                  &lfm[(mi_row >> 3) * lfm_stride + (mi_col >> 3)]);
        FILE *o = fopen(argv[2], "wb");
        if (!o) return 2;
        fwrite(lfm, sizeof *lfm, (size_t)sb_rows * lfm_stride, o);
        fclose(o);
        return 0;
    }
    if (h[0] == 0x504C5653) { /* 'SVLP' */
        fclose(f);
        FILE *o = fopen(argv[2], "wb");
        if (!o) return 2;
        VP9_COMP *cpi = (VP9_COMP *)calloc(1, sizeof *cpi);
        VP9_COMMON *cm = &cpi->common;
        for (int sharp = 0; sharp < 8; sharp++) {
            memset(&cm->lf_info, 0xA5, sizeof cm->lf_info);
            cm->lf.sharpness_level = sharp;
            eb_vp9_loop_filter_init(cm);
            uint8_t out[3][64];
            for (int l = 0; l < 64; l++) {
                const loop_filter_thresh *t = &cm->lf_info.lfthr[l];
                for (int k = 1; k < SIMD_WIDTH; k++)
                    if (t->mblim[k] != t->mblim[0] || t->lim[k] != t->lim[0] || t->hev_thr[k] != t->hev_thr[0]) return 5;
                out[0][l] = t->mblim[0]; out[1][l] = t->lim[0]; out[2][l] = t->hev_thr[0];
            }
            fwrite(out, 1, sizeof out, o);
        }
        cm->bit_depth = VPX_BITS_8;
        for (int key = 0; key < 2; key++)
            for (int q = 0; q < 256; q++) {
                cm->frame_type = key ? KEY_FRAME : INTER_FRAME;
                cm->base_qindex = q;
                cm->lf.filter_level = -1;
                cm->lf.sharpness_level = 3;
                eb_vp9_pick_filter_level(cpi, LPF_PICK_FROM_Q);
                int32_t rec[3] = {eb_vp9_ac_quant(q, 0, VPX_BITS_8), cm->lf.filter_level, cm->lf.sharpness_level};
                fwrite(rec, sizeof(int32_t), 3, o);
            }
        fclose(o);
        return 0;
    }
    if (rd(f, h + 5, 5 * sizeof(int32_t)) || h[0] != 0x464C5653) return 3;
    const int W = h[1], H = h[2], ys = h[3], uvs = h[4], mi_rows = h[5], mi_cols = h[6], lfm_stride = h[7], n_lfm = h[8], y_only = h[9] & 1;
    (void)W;
    svt_lf_thresh thr;
    if (rd(f, &thr, sizeof thr)) return 3;
    if (sizeof(LOOP_FILTER_MASK) != sizeof(svt_lf_mask)) { fprintf(stderr, "LOOP_FILTER_MASK size mismatch\n"); return 4; }
    LOOP_FILTER_MASK *lfm = (LOOP_FILTER_MASK *)calloc((size_t)n_lfm, sizeof *lfm);
    if (rd(f, lfm, sizeof(*lfm) * (size_t)n_lfm)) return 3;
    const size_t ysz = (size_t)ys * H, uvsz = (size_t)uvs * (H / 2);
    uint8_t *y = (uint8_t *)malloc(ysz + 64), *u = (uint8_t *)malloc(uvsz + 64), *v = (uint8_t *)malloc(uvsz + 64);
    if (rd(f, y, ysz) || rd(f, u, uvsz) || rd(f, v, uvsz)) return 3;
    fclose(f);

    setup_rtcd_internal(0);      /* the reference's own dispatch set-up, C kernels */
    setup_rtcd_internal_vp9(0);

    VP9_COMMON  *cm = (VP9_COMMON *)calloc(1, sizeof *cm);
    MACROBLOCKD *xd = (MACROBLOCKD *)calloc(1, sizeof *xd);
    cm->mi_rows = mi_rows;
    cm->mi_cols = mi_cols;
    cm->lf.lfm = lfm;
    cm->lf.lfm_stride = lfm_stride;
    /* thresholds: derived by the reference itself (eb_vp9_loop_filter_init, :250) from the sharpness level carried in
       bits 8..15 of the y_only word; the table in the request is only cross-checked against it */
    cm->lf.sharpness_level = (h[9] >> 8) & 0xff;
    eb_vp9_loop_filter_init(cm);
    for (int l = 0; l < 64; l++)
        if (cm->lf_info.lfthr[l].mblim[0] != thr.mblim[l] || cm->lf_info.lfthr[l].lim[0] != thr.lim[l] ||
            cm->lf_info.lfthr[l].hev_thr[0] != thr.hev_thr[l]) {
            fprintf(stderr, "request thresholds differ from eb_vp9_loop_filter_init at level %d\n", l);
            return 6;
        }
    xd->plane[0].dst.buf = y; xd->plane[0].dst.stride = ys; xd->plane[0].subsampling_x = 0; xd->plane[0].subsampling_y = 0;
    xd->plane[1].dst.buf = u; xd->plane[1].dst.stride = uvs; xd->plane[1].subsampling_x = 1; xd->plane[1].subsampling_y = 1;
    xd->plane[2].dst.buf = v; xd->plane[2].dst.stride = uvs; xd->plane[2].subsampling_x = 1; xd->plane[2].subsampling_y = 1;
    eb_vp9_loop_filter_frame(cm, xd, 1 /* any non-zero level: per-block levels come from lfl_y */, y_only, 0);

    FILE *o = fopen(argv[2], "wb");
    if (!o) return 2;
    fwrite(y, 1, ysz, o); fwrite(u, 1, uvsz, o); fwrite(v, 1, uvsz, o);
    fclose(o);
    return 0;
}
