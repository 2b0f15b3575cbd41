/*
 * lf_masks.c -- host-side (plain C) construction of the per-SB LOOP_FILTER_MASKs from the mode-info grid.
 *
 * Mirrors eb_vp9_build_mask_frame (VPX/vp9_loopfilter.c:1548-1571).  The reference has two formulations of the same
 * rules: the frame-level tree walk eb_vp9_setup_mask (:901-1040) and the per-block eb_vp9_build_mask (:1587-1689)
 * used when masks are built in line.  This file follows the per-block one: every 8x8 unit that is the top-left unit
 * of its prediction block contributes the block's edges at shift = row*8 + col of its SB.
 *
 * The reference's constant tables (prediction / size / transform masks, uv transform size) are pure geometry; they are
 * computed here from the block dimensions instead of being tabulated.
 */
#include "../../include/svtvp9_hip.h"
#include <string.h>

/* block width / height in 8x8 units (sub-8x8 blocks occupy one unit), eb_vp9_num_8x8_blocks_{wide,high}_lookup */
static void block_units(int sb_type, int *w8, int *h8) {
    static const unsigned char w4[13] = {1, 1, 2, 2, 2, 4, 4, 4, 8, 8, 8, 16, 16}; /* width in 4-sample units */
    static const unsigned char h4[13] = {1, 2, 1, 2, 4, 2, 4, 8, 4, 8, 16, 8, 16};
    *w8 = (w4[sb_type] + 1) >> 1;
    *h8 = (h4[sb_type] + 1) >> 1;
}

/* rectangle of ones, w x h units at the origin of a grid with `cols` units per row */
static uint64_t rect_mask(int w, int h, int cols) {
    uint64_t row = ((uint64_t)1 << w) - 1, m = 0;
    for (int i = 0; i < h; i++) m |= row << (i * cols);
    return m;
}

/* units of a 64x64 area whose left (above) side is a transform edge: every unit for 4x4/8x8, every 2nd column (row)
 * for 16x16, every 4th for 32x32 */
static uint64_t tx_edge_mask(int tx_size, int vertical_edges, int cols, int rows) {
    const int step = tx_size <= 1 ? 1 : tx_size == 2 ? 2 : 4;
    uint64_t  m = 0;
    for (int r = 0; r < rows; r++)
        for (int c = 0; c < cols; c++)
            if ((vertical_edges ? c : r) % step == 0) m |= (uint64_t)1 << (r * cols + c);
    return m;
}

/* transform size of the 4:2:0 chroma block: the luma size, capped by the largest transform that fits the chroma
 * block (eb_vp9_uv_txsize_lookup[bsize][tx][1][1]) */
static int uv_tx_size(int sb_type, int tx_size_y) {
    static const unsigned char w4[13] = {1, 1, 2, 2, 2, 4, 4, 4, 8, 8, 8, 16, 16};
    static const unsigned char h4[13] = {1, 2, 1, 2, 4, 2, 4, 8, 4, 8, 16, 8, 16};
    int cw = w4[sb_type] >> 1, ch = h4[sb_type] >> 1; /* chroma size in 4-sample units */
    if (cw < 1) cw = 1;
    if (ch < 1) ch = 1;
    const int m = cw < ch ? cw : ch;
    const int cap = m >= 8 ? 3 : m >= 4 ? 2 : m >= 2 ? 1 : 0;
    return tx_size_y < cap ? tx_size_y : cap;
}

int32_t svt_hip_lf_build_masks(const svt_lf_mode_info *mi, int32_t mi_stride, int32_t mi_rows, int32_t mi_cols,
                               svt_lf_mask *lfm, int32_t lfm_stride) {
    if (!mi || !lfm || mi_rows < 1 || mi_cols < 1 || mi_stride < mi_cols || lfm_stride < (mi_cols + 7) / 8) return SVT_HIP_ERR_BAD_PARAMETER;
    for (int sb_r = 0; sb_r < (mi_rows + 7) / 8; sb_r++)
        for (int sb_c = 0; sb_c < (mi_cols + 7) / 8; sb_c++) {
            svt_lf_mask *m = &lfm[sb_r * lfm_stride + sb_c];
            memset(m, 0, sizeof *m);
            for (int r = 0; r < 8 && sb_r * 8 + r < mi_rows; r++)
                for (int c = 0; c < 8 && sb_c * 8 + c < mi_cols; c++) {
                    const svt_lf_mode_info *b = &mi[(sb_r * 8 + r) * mi_stride + sb_c * 8 + c];
                    if (b->sb_type > 12 || b->tx_size > 3) return SVT_HIP_ERR_BAD_PARAMETER;
                    int w8, h8;
                    block_units(b->sb_type, &w8, &h8);
                    if ((r % h8) != 0 || (c % w8) != 0) continue; /* not the first unit of its block */
                    if (!b->filter_level) continue;               /* level 0: the block is not filtered */
                    const int shift_y = r * 8 + c, shift_uv = (r >> 1) * 4 + (c >> 1);
                    const int with_uv = !(r & 1) && !(c & 1);     /* first 8x8 of a 16x16 area carries the chroma edges */
                    const int txy = b->tx_size, txuv = uv_tx_size(b->sb_type, txy);
                    const int wuv = (w8 + 1) >> 1, huv = (h8 + 1) >> 1;
                    for (int i = 0; i < h8 && r + i < 8; i++)
                        for (int j = 0; j < w8 && c + j < 8; j++) m->lfl_y[shift_y + i * 8 + j] = b->filter_level;
                    /* prediction block edges */
                    m->above_y[txy] |= rect_mask(w8, 1, 8) << shift_y;
                    m->left_y[txy] |= rect_mask(1, h8, 8) << shift_y;
                    if (with_uv) {
                        m->above_uv[txuv] |= (uint16_t)(rect_mask(wuv, 1, 4) << shift_uv);
                        m->left_uv[txuv] |= (uint16_t)(rect_mask(1, huv, 4) << shift_uv);
                    }
                    if (b->skip && b->is_inter) continue; /* no residual, inter: only the block's own border */
                    /* transform edges inside the block, and the inner 4x4 edges */
                    m->above_y[txy] |= (rect_mask(w8, h8, 8) & tx_edge_mask(txy, 0, 8, 8)) << shift_y;
                    m->left_y[txy] |= (rect_mask(w8, h8, 8) & tx_edge_mask(txy, 1, 8, 8)) << shift_y;
                    if (txy == 0) m->int_4x4_y |= rect_mask(w8, h8, 8) << shift_y;
                    if (with_uv) {
                        m->above_uv[txuv] |= (uint16_t)((rect_mask(wuv, huv, 4) & tx_edge_mask(txuv, 0, 4, 4)) << shift_uv);
                        m->left_uv[txuv] |= (uint16_t)((rect_mask(wuv, huv, 4) & tx_edge_mask(txuv, 1, 4, 4)) << shift_uv);
                        if (txuv == 0) m->int_4x4_uv |= (uint16_t)(rect_mask(wuv, huv, 4) << shift_uv);
                    }
                }
        }
    return SVT_HIP_OK;
}
