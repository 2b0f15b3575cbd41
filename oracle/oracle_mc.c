/*
 * oracle_mc.c -- CPU restatement (TEST INFRASTRUCTURE ONLY) of the reference's inter prediction of a picture.
 *   inter_prediction                      Source/Lib/Codec/EbIntraPrediction.c:49-72   (one call per block and plane)
 *   build_inter_predictors                Source/Lib/VPX/vp9_reconinter.c:102-252
 *   eb_vp9_clamp_mv_to_umv_border_sb      Source/Lib/VPX/vp9_reconinter.c:72-92, clamp_mv VPX/vp9_mv.h
 *   inter_predictor                       Source/Lib/VPX/vp9_reconinter.h:23-28  (table set up in VPX/vp9_scale.c:76-84,126-128)
 *   convolve_horiz / _vert / avg forms,
 *   eb_vp9_convolve8_c, _copy_c, _avg_c   Source/Lib/VPX/vpx_convolve.c:20-215
 *   regular 8-tap kernel                  Source/Lib/VPX/vp9_filter.c:32-47 (= the EIGHTTAP filter of the VP9 bitstream
 *                                         specification; eb_vp9_filter_kernels[0] is hard-wired, vp9_reconinter.c:107-109)
 *   mb_to_*_edge                          Source/Lib/Codec/EbEncDecProcess.c:3708-3719
 * Pinned against the reference's own inter_prediction() built from source (oracle/_ref/ref_mc_frame,
 * tests/test_oracle_vs_ref.py) and against tests/golden/mc_reference.npz produced by it.
 */
#include <string.h>
#include "oracle.h"

/* VP9 specification, section "motion vector prediction / interpolation": subpel_filters[REGULAR][16][8] */
static const int16_t k_taps[16][8] = {
    {0, 0, 0, 128, 0, 0, 0, 0},      {0, 1, -5, 126, 8, -3, 1, 0},    {-1, 3, -10, 122, 18, -6, 2, 0},  {-1, 4, -13, 118, 27, -9, 3, -1},
    {-1, 4, -16, 112, 37, -11, 4, -1}, {-1, 5, -18, 105, 48, -14, 4, -1}, {-1, 5, -19, 97, 58, -16, 5, -1}, {-1, 6, -19, 88, 68, -18, 5, -1},
    {-1, 6, -19, 78, 78, -19, 6, -1}, {-1, 5, -18, 68, 88, -19, 6, -1}, {-1, 5, -16, 58, 97, -19, 5, -1}, {-1, 4, -14, 48, 105, -18, 5, -1},
    {-1, 4, -11, 37, 112, -16, 4, -1}, {-1, 3, -9, 27, 118, -13, 4, -1}, {0, 2, -6, 18, 122, -10, 3, -1},  {0, 1, -3, 8, 126, -5, 1, 0}};

static uint8_t clip8(int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }
static int     round7(int sum) { return (sum + 64) >> 7; } /* ROUND_POWER_OF_TWO(sum, FILTER_BITS) */

/* horizontal 8-tap of `rows` rows starting at src (the tap window starts 3 samples to the left), vpx_convolve.c:20-37 */
static void filt_h(const uint8_t *src, ptrdiff_t ss, uint8_t *dst, ptrdiff_t ds, int sx, int w, int rows) {
    const int16_t *t = k_taps[sx];
    for (int y = 0; y < rows; y++, src += ss, dst += ds)
        for (int x = 0; x < w; x++) {
            int sum = 0;
            for (int k = 0; k < 8; k++) sum += src[x - 3 + k] * t[k];
            dst[x] = clip8(round7(sum));
        }
}
/* vertical 8-tap (window starts 3 rows above), vpx_convolve.c:58-76 */
static void filt_v(const uint8_t *src, ptrdiff_t ss, uint8_t *dst, ptrdiff_t ds, int sy, int w, int h) {
    const int16_t *t = k_taps[sy];
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int sum = 0;
            for (int k = 0; k < 8; k++) sum += src[(y - 3 + k) * ss + x] * t[k];
            dst[y * ds + x] = clip8(round7(sum));
        }
}

/* sf->predict[sx != 0][sy != 0][ref]: the first reference is written, the second is averaged into it with
 * ROUND_POWER_OF_TWO(dst + p, 1) -- for every combination the _avg form equals "predict into a temporary, then
 * eb_vp9_convolve_avg_c" (vpx_convolve.c:39-56, 78-97, 171-181, 201-215) */
static void predict_block(const uint8_t *pre, ptrdiff_t ps, uint8_t *dst, ptrdiff_t ds, int sx, int sy, int w, int h, int second) {
    uint8_t tmp[64 * 64], mid[64 * (64 + 7)];
    if (!sx && !sy) {
        for (int y = 0; y < h; y++) memcpy(tmp + 64 * y, pre + y * ps, (size_t)w); /* eb_vp9_convolve_copy_c */
    } else if (sx && !sy) {
        filt_h(pre, ps, tmp, 64, sx, w, h);
    } else if (!sx && sy) {
        filt_v(pre, ps, tmp, 64, sy, w, h);
    } else { /* eb_vp9_convolve8_c: h + 7 rows filtered horizontally into a uint8 buffer, then vertically */
        filt_h(pre - 3 * ps, ps, mid, 64, sx, w, h + 7);
        filt_v(mid + 3 * 64, 64, tmp, 64, sy, w, h);
    }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) dst[y * ds + x] = second ? (uint8_t)((dst[y * ds + x] + tmp[64 * y + x] + 1) >> 1) : tmp[64 * y + x];
}

static int clampi(int v, int lo, int hi) { return v < lo ? lo : v > hi ? hi : v; }

int32_t svt_oracle_inter_pred_frame(const svt_mc_mode_info *mi, int32_t mi_stride, int32_t mi_rows, int32_t mi_cols,
                                    const svt_mc_host_ref ref[2], int32_t use_subpel, uint8_t *pred_y, uint8_t *pred_u,
                                    uint8_t *pred_v) {
    const int W = mi_cols * 8;
    for (int r = 0; r < mi_rows; r++)
        for (int c = 0; c < mi_cols; c++) {
            const svt_mc_mode_info *m = &mi[r * mi_stride + c];
            if (m->ref_list[0] < 0 || m->bw8 < 1 || m->bh8 < 1) continue;
            if ((r % m->bh8) || (c % m->bw8)) continue; /* not the block's first unit */
            const int bx = c * 8, by = r * 8;             /* context_ptr->block_origin_x / y */
            /* EbEncDecProcess.c:3708-3719 */
            const int to_left = -(c * 8 * 8), to_right = (mi_cols - m->bw8 - c) * 8 * 8;
            const int to_top = -(r * 8 * 8), to_bottom = (mi_rows - m->bh8 - r) * 8 * 8;
            for (int plane = 0; plane < 3; plane++) {
                const int ss = plane ? 1 : 0;
                const int bw = (m->bw8 * 8) >> ss, bh = (m->bh8 * 8) >> ss;
                uint8_t  *dst = plane == 0 ? pred_y + (size_t)by * W + bx
                                           : (plane == 1 ? pred_u : pred_v) + (size_t)(by >> 1) * (W >> 1) + (bx >> 1);
                const ptrdiff_t ds = plane ? W >> 1 : W;
                const int nref = m->ref_list[1] >= 0 ? 2 : 1;
                for (int k = 0; k < nref; k++) {
                    const svt_mc_host_ref *rf = &ref[m->ref_list[k] ? 1 : 0];
                    /* eb_vp9_clamp_mv_to_umv_border_sb: result in 1/16 sample of this plane */
                    const int spel_left = (4 + bw) << 4, spel_right = spel_left - 16;
                    const int spel_top = (4 + bh) << 4, spel_bottom = spel_top - 16;
                    const int sc = 1 << (1 - ss);
                    int mv_row = m->mv_row[k] * sc, mv_col = m->mv_col[k] * sc;
                    mv_col = clampi(mv_col, to_left * sc - spel_left, to_right * sc + spel_right);
                    mv_row = clampi(mv_row, to_top * sc - spel_top, to_bottom * sc + spel_bottom);
                    /* vp9_reconinter.c:150-166: chroma positions use ROUND_UV() = multiples of 8 luma samples (always true here) */
                    const ptrdiff_t ps = plane ? rf->uv_stride : rf->y_stride;
                    const uint8_t  *pre = plane == 0 ? rf->y + rf->org_x + bx + (ptrdiff_t)(rf->org_y + by) * ps
                                                     : (plane == 1 ? rf->u : rf->v) + ((rf->org_x + ((bx >> 3) << 3)) >> 1) +
                                                           (ptrdiff_t)((rf->org_y + ((by >> 3) << 3)) >> 1) * ps;
                    int s_row, s_col, sx, sy;
                    if (use_subpel) { /* :170-174 */
                        s_row = mv_row; s_col = mv_col; sx = s_col & 15; sy = s_row & 15;
                    } else if (plane) { /* :176-180 -- [quirk] multiples of 8 are kept but only 3 bits are tested */
                        s_row = (mv_row + 4) & ~7; s_col = (mv_col + 4) & ~7; sx = s_col & 7; sy = s_row & 7;
                    } else {            /* :181-186 */
                        s_row = (mv_row + 8) & ~15; s_col = (mv_col + 8) & ~15; sx = s_col & 15; sy = s_row & 15;
                    }
                    pre += (s_row >> 4) * ps + (s_col >> 4);
                    predict_block(pre, ps, dst, ds, sx, sy, bw, bh, k);
                }
            }
        }
    return 0;
}
