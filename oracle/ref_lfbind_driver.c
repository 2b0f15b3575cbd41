/*
 * ref_lfbind_driver.c -- harness for the deblocking call site of the reference's encode pass (Source/Lib/Codec/EbEncDecProcess.c:5676-5686):
 *     eb_vp9_build_mask_frame(&cpi->common, lf->filter_level, 0);  eb_vp9_loop_filter_frame(&cpi->common, e_mbd, lf->filter_level, 0, 0);
 * run by the reference's own code on a real VP9_COMMON (mode-info grid, loop-filter state initialised by eb_vp9_loop_filter_init, masks built
 * by eb_vp9_build_mask_frame) and MACROBLOCKD (plane destinations as the call site sets them, :5658-5674), and then with the second call
 * replaced by the BINDING of this repository (integration/loop_filter_binding.h -> svt_hip_lf_frame) on an identical copy of the frame.
 * TEST INFRASTRUCTURE ONLY (rules: ref_me_driver.c; RTCD set-up as ref_lf_driver.c).
 *
 * request : int32 magic 'SVLB', width, height (luma, multiples of 8), y_stride, uv_stride, rows_y, rows_uv (allocated rows of the planes),
 *           filter_level, sharpness, y_only, run_binding, device; mi_rows * mi_cols cells of 6 bytes {sb_type, tx_size, skip, ref_frame[0],
 *           mode, segment_id}; Y (y_stride * rows_y), U, V (uv_stride * rows_uv)
 * response: the three planes as the reference left them, int32 binding_rc, the three planes as the binding left them (when run_binding),
 *           then the LOOP_FILTER_MASK array eb_vp9_build_mask_frame built ([sb_rows][lfm_stride]), then one double: the seconds the
 *           reference's two calls took (clock_gettime around them: bench.py's cpu_baseline.reference_lf)
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <time.h>

#define RTCD_C
#include "vpx_dsp_rtcd.h"
#include "vp9_rtcd.h"
#include "vp9_onyxc_int.h"
#include "vp9_blockd.h"
#include "vp9_loopfilter.h"

#include "../include/svtvp9_hip.h"
#include "../integration/loop_filter_binding.h"

uint32_t eb_vp9_ASM_TYPES = 0;

static int rd(FILE *f, void *p, size_t n) { return fread(p, 1, n, f) == n ? 0 : -1; }

int main(int argc, char **argv) {
    if (argc < 3) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    int32_t h[13];
    if (rd(f, h, sizeof h) || h[0] != 0x424C5653) return 3; /* 'SVLB' */
    const int W = h[1], H = h[2], ys = h[3], uvs = h[4], rows_y = h[5], rows_uv = h[6], level = h[7], sharp = h[8], y_only = h[9], run_binding = h[10], device = h[11];
    const int mi_rows = H / 8, mi_cols = W / 8, sb_rows = (mi_rows + 7) / 8, lfm_stride = (mi_cols + 7) / 8;
    const size_t n = (size_t)mi_rows * mi_cols;
    uint8_t     *cells = (uint8_t *)malloc(n * 6);
    if (rd(f, cells, n * 6)) return 3;
    const size_t ysz = (size_t)ys * rows_y, uvsz = (size_t)uvs * rows_uv;
    uint8_t     *src[3] = {(uint8_t *)malloc(ysz + 64), (uint8_t *)malloc(uvsz + 64), (uint8_t *)malloc(uvsz + 64)};
    if (rd(f, src[0], ysz) || rd(f, src[1], uvsz) || rd(f, src[2], uvsz)) return 3;
    fclose(f);

    setup_rtcd_internal(0); /* the reference's own dispatch set-up, C kernels (`-asm 0`) */
    setup_rtcd_internal_vp9(0);

    VP9_COMMON *cm = (VP9_COMMON *)calloc(1, sizeof *cm);
    cm->mi_rows = mi_rows; cm->mi_cols = mi_cols; cm->mi_stride = mi_cols;
    ModeInfo  *mis  = (ModeInfo *)calloc(n, sizeof *mis);
    ModeInfo **grid = (ModeInfo **)calloc(n, sizeof *grid);
    for (size_t i = 0; i < n; i++) {
        mis[i].sb_type = (BLOCK_SIZE)cells[6 * i]; mis[i].tx_size = (TX_SIZE)cells[6 * i + 1]; mis[i].skip = cells[6 * i + 2];
        mis[i].ref_frame[0] = (MV_REFERENCE_FRAME)cells[6 * i + 3]; mis[i].mode = (PREDICTION_MODE)cells[6 * i + 4];
        mis[i].segment_id = cells[6 * i + 5];
        grid[i] = &mis[i];
    }
    cm->mi_grid_visible = grid;
    cm->lf.lfm = (LOOP_FILTER_MASK *)calloc((size_t)sb_rows * lfm_stride, sizeof(LOOP_FILTER_MASK));
    cm->lf.lfm_stride = lfm_stride;
    cm->lf.sharpness_level = sharp;
    eb_vp9_loop_filter_init(cm); /* as the encoder's set-up: threshold tables for the sharpness, last_sharpness_level */
    cm->lf.filter_level = level;

    /* the call site: masks, then the filter -- reference */
    struct timespec t0, t1, t2;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    eb_vp9_build_mask_frame(cm, level, 0);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    /* (the filter adjusts the masks in place, eb_vp9_adjust_mask :786: the binding, which stands where the filter call stands, gets them as
       eb_vp9_build_mask_frame left them) */
    const size_t      lfm_bytes = sizeof(LOOP_FILTER_MASK) * (size_t)sb_rows * lfm_stride;
    LOOP_FILTER_MASK *lfm_built = (LOOP_FILTER_MASK *)malloc(lfm_bytes);
    memcpy(lfm_built, cm->lf.lfm, lfm_bytes);
    uint8_t *a[3], *b[3];
    for (int k = 0; k < 3; k++) {
        const size_t sz = k ? uvsz : ysz;
        a[k] = (uint8_t *)malloc(sz + 64); b[k] = (uint8_t *)malloc(sz + 64);
        memcpy(a[k], src[k], sz); memcpy(b[k], src[k], sz);
    }
    MACROBLOCKD *xd = (MACROBLOCKD *)calloc(1, sizeof *xd);
    for (int k = 0; k < 3; k++) { xd->plane[k].dst.buf = a[k]; xd->plane[k].dst.stride = k ? uvs : ys; xd->plane[k].subsampling_x = xd->plane[k].subsampling_y = k ? 1 : 0; }
    clock_gettime(CLOCK_MONOTONIC, &t2);
    eb_vp9_loop_filter_frame(cm, xd, level, y_only, 0);
    {
        struct timespec t3;
        clock_gettime(CLOCK_MONOTONIC, &t3);
        t2.tv_sec = t3.tv_sec - t2.tv_sec; t2.tv_nsec = t3.tv_nsec - t2.tv_nsec; /* the filter alone */
    }
    const double seconds = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec) + (double)t2.tv_sec + 1e-9 * (double)t2.tv_nsec;

    FILE *o = fopen(argv[2], "wb");
    if (!o) return 2;
    fwrite(a[0], 1, ysz, o); fwrite(a[1], 1, uvsz, o); fwrite(a[2], 1, uvsz, o);
    /* the same call site with the binding in place of eb_vp9_loop_filter_frame */
    int32_t brc = -100;
    if (run_binding) {
        svt_hip_ctx *hip = NULL;
        for (int k = 0; k < 3; k++) xd->plane[k].dst.buf = b[k];
        memcpy(cm->lf.lfm, lfm_built, lfm_bytes);
        if (svt_hip_ctx_create(&hip, device) != 0) { fprintf(stderr, "binding: %s\n", svt_hip_last_error()); brc = -101; }
        else {
            brc = svt_hip_bind_loop_filter_frame(hip, cm, xd, level, y_only, 0);
            if (brc) fprintf(stderr, "binding: %s\n", svt_hip_last_error());
            svt_hip_ctx_destroy(hip);
        }
        fwrite(&brc, 4, 1, o);
        fwrite(b[0], 1, ysz, o); fwrite(b[1], 1, uvsz, o); fwrite(b[2], 1, uvsz, o);
    } else fwrite(&brc, 4, 1, o);
    fwrite(lfm_built, 1, lfm_bytes, o);
    fwrite(&seconds, sizeof seconds, 1, o);
    fclose(o);
    return 0;
}
