/*
 * encdec_core.h -- the per-superblock rules of the picture-level EncDec driver (csrc/encdec.hip), written once as plain inline
 * functions that compile both for the device (hipcc: the kernels call them) and for the host (gcc: host/lf_masks.c, host/encdec_host.c
 * call the very same text, which is what the CPU tests pin against the reference).  No tables in memory: block geometry is
 * arithmetic on packed constants, so there is nothing to stage and nothing that differs between the two compilations.
 *
 *   svt_lf_mask_build_sb     one LOOP_FILTER_MASK from the mode-info grid = eb_vp9_build_mask_frame, per-block formulation
 *                            eb_vp9_build_mask (VPX/vp9_loopfilter.c:1548-1571, 1587-1689)
 *   svt_tq_unit_counts/_emit the transform blocks perform_coding_loop is called for in the encode pass of one prediction block
 *                            (encode_pass_sb, Codec/EbEncDecProcess.c:3813-3960: luma TUs of the block's transform size, then Cb,
 *                            then Cr with uv_txsize_lookup), as svt_tq_block descriptors
 *   svt_md_default_unit      NOT the reference's mode decision (out of scope): a deterministic stand-in that turns the ME results of
 *                            an SB into a valid mode-info grid so that the stages behind mode decision can run from the public API
 *                            without a host-supplied decision
 */
#ifndef SVT_ENCDEC_CORE_H
#define SVT_ENCDEC_CORE_H

#include <stdint.h>
#include "../../include/svtvp9_hip.h"

#if defined(__HIPCC__)
#define SVT_HD __host__ __device__ static inline
#else
#define SVT_HD static inline
#endif

/* ------------------------------------------------------------------------------------------------------------------------ */
/* block geometry (VPX/vp9_common_data.c: b_width_log2_lookup / b_height_log2_lookup, one nibble per BLOCK_SIZE 0..12)       */
/* ------------------------------------------------------------------------------------------------------------------------ */
SVT_HD int svt_blk_w4log2(int sb_type) { return (int)((0x4433322211100ull >> (4 * sb_type)) & 15); } /* width  = 4 << . */
SVT_HD int svt_blk_h4log2(int sb_type) { return (int)((0x4343232121010ull >> (4 * sb_type)) & 15); } /* height = 4 << . */
/* width / height in 8x8 units; a sub-8x8 block occupies one unit (eb_vp9_num_8x8_blocks_{wide,high}_lookup) */
SVT_HD int svt_blk_w8(int sb_type) { const int l = svt_blk_w4log2(sb_type); return l ? 1 << (l - 1) : 1; }
SVT_HD int svt_blk_h8(int sb_type) { const int l = svt_blk_h4log2(sb_type); return l ? 1 << (l - 1) : 1; }

/* transform size of the 4:2:0 chroma block: the luma size, capped by the largest transform that fits the chroma block
 * (eb_vp9_uv_txsize_lookup[bsize][tx][1][1], VPX/vp9_common_data.c) */
SVT_HD int svt_uv_tx_size(int sb_type, int tx_size_y) {
    int cw = (1 << svt_blk_w4log2(sb_type)) >> 1, ch = (1 << svt_blk_h4log2(sb_type)) >> 1; /* chroma size in 4-sample units */
    if (cw < 1) cw = 1;
    if (ch < 1) ch = 1;
    const int m = cw < ch ? cw : ch;
    const int cap = m >= 8 ? 3 : m >= 4 ? 2 : m >= 2 ? 1 : 0;
    return tx_size_y < cap ? tx_size_y : cap;
}

/* ------------------------------------------------------------------------------------------------------------------------ */
/* LOOP_FILTER_MASK of one SB                                                                                                 */
/* ------------------------------------------------------------------------------------------------------------------------ */
/* rectangle of ones, w x h units at the origin of a grid with `cols` (8 or 4) units per row: the row pattern replicated into the
 * first h rows by one multiplication */
SVT_HD uint64_t svt_rect_mask(int w, int h, int cols) {
    const uint64_t row = ((uint64_t)1 << w) - 1;
    const uint64_t rep = cols == 8 ? 0x0101010101010101ull : 0x1111ull;
    const int      bits = h * cols;
    return (row * rep) & (bits >= 64 ? ~(uint64_t)0 : (((uint64_t)1 << bits) - 1));
}
/* units of an 8 x 8 (luma) or 4 x 4 (chroma) area whose left (vertical_edges) / above side is a transform edge: every unit for
 * 4x4 / 8x8 transforms, every 2nd column (row) for 16x16, every 4th for 32x32 */
SVT_HD uint64_t svt_tx_edge_mask(int tx_size, int vertical_edges, int cols, int rows) {
    (void)rows;
    if (cols == 8) {
        if (tx_size <= 1) return ~(uint64_t)0;
        if (tx_size == 2) return vertical_edges ? 0x5555555555555555ull : 0x00FF00FF00FF00FFull;
        return vertical_edges ? 0x1111111111111111ull : 0x000000FF000000FFull;
    }
    if (tx_size <= 1) return 0xFFFFull;
    if (tx_size == 2) return vertical_edges ? 0x5555ull : 0x0F0Full;
    return vertical_edges ? 0x1111ull : 0x000Full;
}

/* What ONE 8x8 unit adds to the LOOP_FILTER_MASK of its SB: the unit that is the first of its prediction block contributes the block's
 * edges at shift = row * 8 + col (the reference's per-block eb_vp9_build_mask); every unit carries the filter level of the block that
 * covers it.  The host form ORs the 64 contributions of an SB one after the other, the device form one per lane with a wave-wide OR --
 * the contribution itself is this one text. */
typedef struct svt_lf_unit_masks {
    uint64_t left_y[4], above_y[4], int_4x4_y;
    uint16_t left_uv[4], above_uv[4], int_4x4_uv;
    uint8_t  level;  /* lfl_y of this unit */
    uint8_t  bad;    /* malformed record */
} svt_lf_unit_masks;

SVT_HD void svt_lf_mask_unit(const svt_lf_mode_info *mi, int mi_stride, int mi_rows, int mi_cols, int sb_r, int sb_c, int r, int c, svt_lf_unit_masks *m) {
    for (int i = 0; i < 4; i++) { m->left_y[i] = m->above_y[i] = 0; m->left_uv[i] = m->above_uv[i] = 0; }
    m->int_4x4_y = 0; m->int_4x4_uv = 0; m->level = 0; m->bad = 0;
    if (sb_r * 8 + r >= mi_rows || sb_c * 8 + c >= mi_cols) {
        /* a unit beyond the picture edge has no record, but a block that starts inside the picture may reach over it: the reference
           writes that block's level over its whole w x h span (eb_vp9_build_mask, VPX/vp9_loopfilter.c:1611-1616) */
        for (int kh = 8; kh >= 1; kh >>= 1)
            for (int kw = 8; kw >= 1; kw >>= 1) {
                const int oy = r - r % kh, ox = c - c % kw;
                if (sb_r * 8 + oy >= mi_rows || sb_c * 8 + ox >= mi_cols) continue;
                const svt_lf_mode_info *o = &mi[(sb_r * 8 + oy) * mi_stride + sb_c * 8 + ox];
                if (o->sb_type > 12) continue;
                if (svt_blk_h8(o->sb_type) == kh && svt_blk_w8(o->sb_type) == kw) { m->level = o->filter_level; return; }
            }
        return;
    }
    const svt_lf_mode_info *b = &mi[(sb_r * 8 + r) * mi_stride + sb_c * 8 + c];
    if (b->sb_type > 12 || b->tx_size > 3) { m->bad = 1; return; }
    const int w8 = svt_blk_w8(b->sb_type), h8 = svt_blk_h8(b->sb_type);
    {   /* the level of the covering block: its first unit's record (every unit of a block carries the block's values; a block that
           starts outside this SB does not exist -- blocks are aligned to their size) */
        const svt_lf_mode_info *o = &mi[(sb_r * 8 + r - r % h8) * mi_stride + sb_c * 8 + c - c % w8];
        m->level = o->filter_level;
    }
    if ((r % h8) != 0 || (c % w8) != 0) return; /* not the first unit of its block */
    if (!b->filter_level) return;               /* level 0: the block is not filtered */
    const int shift_y = r * 8 + c, shift_uv = (r >> 1) * 4 + (c >> 1);
    const int with_uv = !(r & 1) && !(c & 1);   /* first 8x8 of a 16x16 area carries the chroma edges */
    const int txy = b->tx_size, txuv = svt_uv_tx_size(b->sb_type, txy);
    const int wuv = (w8 + 1) >> 1, huv = (h8 + 1) >> 1;
    /* the block's words are gathered in scalars and go to the entry of their transform size with CONSTANT indices at the end: an
       array indexed by a run-time value lives in private memory on the device (the first version of the mask kernel wrote 290 MB of
       scratch per mini-GOP that way) */
    uint64_t ay, ly;
    uint32_t auv = 0, luv = 0;
    /* prediction block edges */
    ay = svt_rect_mask(w8, 1, 8) << shift_y;
    ly = svt_rect_mask(1, h8, 8) << shift_y;
    if (with_uv) {
        auv = (uint32_t)(svt_rect_mask(wuv, 1, 4) << shift_uv);
        luv = (uint32_t)(svt_rect_mask(1, huv, 4) << shift_uv);
    }
    if (!(b->skip && b->is_inter)) { /* (no residual, inter: only the block's own border) */
        /* transform edges inside the block, and the inner 4x4 edges */
        ay |= (svt_rect_mask(w8, h8, 8) & svt_tx_edge_mask(txy, 0, 8, 8)) << shift_y;
        ly |= (svt_rect_mask(w8, h8, 8) & svt_tx_edge_mask(txy, 1, 8, 8)) << shift_y;
        if (txy == 0) m->int_4x4_y |= svt_rect_mask(w8, h8, 8) << shift_y;
        if (with_uv) {
            auv |= (uint32_t)((svt_rect_mask(wuv, huv, 4) & svt_tx_edge_mask(txuv, 0, 4, 4)) << shift_uv);
            luv |= (uint32_t)((svt_rect_mask(wuv, huv, 4) & svt_tx_edge_mask(txuv, 1, 4, 4)) << shift_uv);
            if (txuv == 0) m->int_4x4_uv |= (uint16_t)(svt_rect_mask(wuv, huv, 4) << shift_uv);
        }
    }
    for (int i = 0; i < 4; i++) {
        m->above_y[i] = i == txy ? ay : 0; m->left_y[i] = i == txy ? ly : 0;
        m->above_uv[i] = (uint16_t)(i == txuv ? auv : 0); m->left_uv[i] = (uint16_t)(i == txuv ? luv : 0);
    }
}

/* host form: one SB.  Returns 0, or -1 when a record is malformed (sb_type > 12 or tx_size > 3). */
SVT_HD int svt_lf_mask_build_sb(const svt_lf_mode_info *mi, int mi_stride, int mi_rows, int mi_cols, int sb_r, int sb_c, svt_lf_mask *m) {
    {
        uint64_t *z = (uint64_t *)m; /* 160 bytes */
        for (int i = 0; i < (int)(sizeof(svt_lf_mask) / 8); i++) z[i] = 0;
    }
    for (int r = 0; r < 8; r++)
        for (int c = 0; c < 8; c++) {
            svt_lf_unit_masks u;
            svt_lf_mask_unit(mi, mi_stride, mi_rows, mi_cols, sb_r, sb_c, r, c, &u);
            if (u.bad) return -1;
            for (int i = 0; i < 4; i++) { m->left_y[i] |= u.left_y[i]; m->above_y[i] |= u.above_y[i]; m->left_uv[i] |= u.left_uv[i]; m->above_uv[i] |= u.above_uv[i]; }
            m->int_4x4_y |= u.int_4x4_y; m->int_4x4_uv |= u.int_4x4_uv;
            m->lfl_y[r * 8 + c] = u.level;
        }
    return 0;
}

/* ------------------------------------------------------------------------------------------------------------------------ */
/* transform blocks of a picture from its mode-info grid                                                                      */
/* ------------------------------------------------------------------------------------------------------------------------ */
/* Coefficients live where the reference keeps them -- per SB, each block's N*N coefficients contiguous (its
 * quantized_coeff_buffer advances by whole blocks, Codec/EbEncDecProcess.c:4100-4108) -- but addressed by POSITION instead of
 * by coding order, so that every consumer can find a block's coefficients without the block list: inside the SB's plane area the
 * 4x4 units are in z-order, which makes every aligned N x N block a contiguous run of N*N elements. */
SVT_HD uint32_t svt_zorder4(int x4, int y4) { /* interleave the bits of two 4-bit numbers: x in the even positions */
    uint32_t v = 0;
    for (int b = 0; b < 4; b++) v |= (uint32_t)((x4 >> b) & 1) << (2 * b) | (uint32_t)((y4 >> b) & 1) << (2 * b + 1);
    return v;
}
/* element offset of the transform block at luma-plane (plane 0) or chroma-plane (1, 2) sample position (x, y) */
SVT_HD uint32_t svt_coeff_offset(int plane, int x, int y, int sb_cols) {
    const int sbw = plane ? 32 : 64;
    const int sb = (y / sbw) * sb_cols + (x / sbw);
    return (uint32_t)sb * SVT_SB_COEFFS + (plane == 0 ? 0u : plane == 1 ? 4096u : 5120u) + svt_zorder4((x % sbw) >> 2, (y % sbw) >> 2) * 16u;
}
/* position code of a transform block: luma transform type << 30 | picture-in-batch << 24 | plane << 22 | (y >> 2) << 11 | (x >> 2), x / y in
 * samples of that plane (pictures are at most 8192 x 4320: 11 bits each; up to 64 pictures per batch).  With the block's transform size
 * (the list it sits in) and the picture's svt_tq_pic_geom it IS the block: svt_tq_block_from_pos rebuilds the 32-byte descriptor, which is
 * what the transform kernels do when they walk the device-built lists -- 4 bytes per block travel instead of 36 (round 5). */
SVT_HD uint32_t svt_tq_pos(int pic, int plane, int x, int y) {
    return (uint32_t)pic << 24 | (uint32_t)plane << 22 | (uint32_t)(y >> 2) << 11 | (uint32_t)(x >> 2);
}
SVT_HD int svt_tq_pos_pic(uint32_t p) { return (int)(p >> 24) & 63; }
/* the descriptor svt_tq_unit_emit writes for the inter block of transform size ts at position code p; g = geometry of picture svt_tq_pos_pic(p) */
SVT_HD void svt_tq_block_from_pos(uint32_t p, int ts, const svt_tq_pic_geom *g, const uint32_t *iscan_off /* [4][4] */, int sb_cols, svt_tq_block *k) {
    const int plane = (int)(p >> 22) & 3, y = (int)((p >> 11) & 0x7ff) << 2, x = (int)(p & 0x7ff) << 2, c = plane ? 1 : 0, tt = (int)(p >> 30);
    k->src_off   = g->src_off[plane] + (uint32_t)y * g->src_stride[c] + (uint32_t)x;
    k->pred_off  = g->pred_off[plane] + (uint32_t)y * g->pred_stride[c] + (uint32_t)x;
    k->recon_off = g->recon_off[plane] + (uint32_t)y * g->recon_stride[c] + (uint32_t)x;
    k->coeff_off = g->coeff_base + svt_coeff_offset(plane, x, y, sb_cols);
    k->iscan_off = iscan_off[ts * 4 + tt];
    k->src_stride = g->src_stride[c]; k->pred_stride = g->pred_stride[c]; k->recon_stride = g->recon_stride[c];
    k->tx_size = (uint8_t)ts; k->tx_type = (uint8_t)tt; k->qtab = (uint8_t)c; k->do_recon = g->do_recon; k->partial32 = 0;
    k->pad_[0] = (uint8_t)(SVT_TQ_RATE_INFO(0, c, 1) | SVT_TQ_RECON_SET(g->recon_set));
}

/* Is unit (ur, uc) the first unit of a well-formed prediction block that lies inside the picture?  1 yes, 0 no (covered by a
 * block that starts elsewhere, or outside the picture), -1 malformed (bad sizes, transform larger than the block, block
 * crossing the picture edge or not aligned to its own size). */
SVT_HD int svt_tq_unit_is_origin(const svt_lf_mode_info *mi, int mi_stride, int mi_rows, int mi_cols, int ur, int uc) {
    if (ur >= mi_rows || uc >= mi_cols) return 0;
    const svt_lf_mode_info *b = &mi[ur * mi_stride + uc];
    if (b->sb_type > 12 || b->tx_size > 3) return -1;
    const int w8 = svt_blk_w8(b->sb_type), h8 = svt_blk_h8(b->sb_type);
    if ((ur % h8) != 0 || (uc % w8) != 0) return 0;
    if (ur + h8 > mi_rows || uc + w8 > mi_cols) return -1;
    const int n8 = b->tx_size == 0 ? 1 : 1 << (b->tx_size - 1); /* transform width in units (4x4: inside one unit) */
    if (n8 > w8 || n8 > h8) return -1;
    if (b->sb_type < 3 && b->tx_size != 0) return -1;
    return 1;
}

/* number of transform blocks per size that the block starting at unit (ur, uc) adds; cnt[4] is ADDED to.  The lists hold the blocks
 * of INTER prediction blocks only: an intra block (is_inter == 0) predicts from its neighbours' reconstruction, so it cannot ride in a
 * size-grouped batch -- the intra pass (intra_kernel.hip) codes it after the batch, when its inter neighbours are reconstructed. */
SVT_HD void svt_tq_unit_counts(const svt_lf_mode_info *mi, int mi_stride, int ur, int uc, int cnt[4]) {
    const svt_lf_mode_info *b = &mi[ur * mi_stride + uc];
    if (!b->is_inter) return;
    const int bw = svt_blk_w8(b->sb_type) * 8, bh = svt_blk_h8(b->sb_type) * 8;
    const int n = 4 << b->tx_size, txuv = svt_uv_tx_size(b->sb_type, b->tx_size), nuv = 4 << txuv;
    cnt[b->tx_size] += (bw / n) * (bh / n);
    cnt[txuv] += 2 * ((bw / 2) / nuv) * ((bh / 2) / nuv);
}

/* Writes the position codes (and, when blocks is not null, the descriptors) of that block: pos / blocks[base[s] ..] for its transform
 * blocks of size s, in the order luma (raster inside the block), Cb, Cr; base[s] is ADVANCED.  iscan_off[tx_size][tx_type] = element offset of the inverse-scan table.  The luma transform
 * type travels in svt_lf_mode_info.pad_[0] (0 = DCT_DCT: every inter block; chroma and 32x32 are always DCT_DCT here). */
SVT_HD void svt_tq_unit_emit(const svt_lf_mode_info *mi, int mi_stride, int ur, int uc, const svt_tq_pic_geom *g, const uint32_t *iscan_off /* [4][4] */,
                             uint32_t base[4], svt_tq_block *blocks, uint32_t *pos) {
    const svt_lf_mode_info *b = &mi[ur * mi_stride + uc];
    if (!b->is_inter) return;
    const int bw = svt_blk_w8(b->sb_type) * 8, bh = svt_blk_h8(b->sb_type) * 8;
    const int sb_cols = (g->width + 63) >> 6;
    for (int plane = 0; plane < 3; plane++) {
        const int ts = plane ? svt_uv_tx_size(b->sb_type, b->tx_size) : b->tx_size, n = 4 << ts;
        const int x0 = plane ? uc * 4 : uc * 8, y0 = plane ? ur * 4 : ur * 8, pw = plane ? bw / 2 : bw, ph = plane ? bh / 2 : bh;
        const int tt = (plane == 0 && ts < 3) ? (b->pad_[0] & 3) : 0;
        for (int y = y0; y < y0 + ph; y += n)
            for (int x = x0; x < x0 + pw; x += n) {
                const uint32_t i = base[ts]++;
                pos[i] = svt_tq_pos(g->pic, plane, x, y) | (uint32_t)tt << 30;
                if (blocks) svt_tq_block_from_pos(pos[i], ts, g, iscan_off, sb_cols, &blocks[i]); /* (null: the consumer works from the position codes) */
            }
    }
}

/* ------------------------------------------------------------------------------------------------------------------------ */
/* stand-in for mode decision                                                                                                 */
/* ------------------------------------------------------------------------------------------------------------------------ */
/* The reference's mode decision (Codec/EbModeDecision*.c, the RD search of EbEncDecProcess.c) is host control logic outside the hot
 * path.  So that the public API can run the stages behind it without a host-supplied decision, this rule turns the ME results of an
 * SB into a partition: bottom-up over the PU tree 16x16 -> 32x32 -> 64x64, a parent replaces its four children when
 *      distortion(parent) <= sum distortion(children) + 3 * lambda          (one motion vector set instead of four)
 * with the best ME candidate's distortion of each PU; blocks that would cross the picture edge are split, down to 8x8 at the
 * edge.  Every block is inter, takes the direction and vectors of its own PU's best candidate (quarter-sample -> 1/8 sample) and
 * the transform of its own size (32x32 for 64x64).  Deterministic, no claim of coding efficiency.
 * Unit (r, c) of the SB at (sb_row, sb_col); res = the SB's 85 ME records. */
SVT_HD uint32_t svt_md_cost(const svt_me_pu_result *res, int pu) { return res[pu].distortion_direction[0].distortion; }

SVT_HD void svt_md_default_unit(const svt_me_pu_result *res, int r, int c, int sb_row, int sb_col, int mi_rows, int mi_cols, uint32_t lambda,
                                int filter_level, svt_mc_mode_info *mc, svt_lf_mode_info *lf) {
    const int ur0 = sb_row * 8, uc0 = sb_col * 8;
    const int q32 = (r >> 2) * 2 + (c >> 2), q16 = ((r >> 1) & 1) * 2 + ((c >> 1) & 1), q8 = (r & 1) * 2 + (c & 1);
    /* which block sizes fit the picture at this unit's position */
    const int fit64 = ur0 + 8 <= mi_rows && uc0 + 8 <= mi_cols;
    const int fit32 = ur0 + (r & ~3) + 4 <= mi_rows && uc0 + (c & ~3) + 4 <= mi_cols;
    const int fit16 = ur0 + (r & ~1) + 2 <= mi_rows && uc0 + (c & ~1) + 2 <= mi_cols;
    /* cost of a 32x32 area at its best: the 32x32 PU or its four 16x16 PUs (areas that cross the edge never merge) */
    uint64_t cost64_children = 0;
    int      merged32_here = 0, all32_fit = 1;
    for (int a = 0; a < 4; a++) {
        const int ar = (a >> 1) * 4, ac = (a & 1) * 4;
        const int fits = ur0 + ar + 4 <= mi_rows && uc0 + ac + 4 <= mi_cols;
        uint64_t  s16 = 0;
        for (int k = 0; k < 4; k++) s16 += svt_md_cost(res, 5 + 4 * a + k);
        const uint64_t p32 = svt_md_cost(res, 1 + a);
        const int      merge = fits && p32 <= s16 + 3ull * lambda;
        cost64_children += merge ? p32 + lambda : s16 + 4ull * lambda;
        all32_fit &= fits;
        if (a == q32) merged32_here = merge;
    }
    const int merge64 = fit64 && all32_fit && (uint64_t)svt_md_cost(res, 0) + lambda <= cost64_children;
    int pu, sb_type, tx, w8;
    if (merge64) { pu = 0; sb_type = 12; tx = 3; w8 = 8; }
    else if (merged32_here && fit32) { pu = 1 + q32; sb_type = 9; tx = 3; w8 = 4; }
    else if (fit16) { pu = 5 + 4 * q32 + q16; sb_type = 6; tx = 2; w8 = 2; }
    else { pu = 21 + 16 * q32 + 4 * q16 + q8; sb_type = 3; tx = 1; w8 = 1; }
    const svt_me_pu_result *p = &res[pu];
    const int d = (int)p->distortion_direction[0].direction; /* 0 list 0, 1 list 1, 2 bi-prediction */
    mc->bw8 = mc->bh8 = (uint8_t)w8;
    mc->ref_list[0] = (int8_t)(d == 1 ? 1 : 0);
    mc->ref_list[1] = (int8_t)(d == 2 ? 1 : -1);
    mc->mv_row[0] = (int16_t)(2 * (d == 1 ? p->y_mv_l1 : p->y_mv_l0));
    mc->mv_col[0] = (int16_t)(2 * (d == 1 ? p->x_mv_l1 : p->x_mv_l0));
    mc->mv_row[1] = (int16_t)(d == 2 ? 2 * p->y_mv_l1 : 0);
    mc->mv_col[1] = (int16_t)(d == 2 ? 2 * p->x_mv_l1 : 0);
    lf->sb_type = (uint8_t)sb_type; lf->tx_size = (uint8_t)tx; lf->skip = 0; lf->is_inter = 1; lf->filter_level = (uint8_t)filter_level;
    lf->pad_[0] = lf->pad_[1] = lf->pad_[2] = 0;
}

#endif /* SVT_ENCDEC_CORE_H */
