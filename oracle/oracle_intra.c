/*
 * oracle_intra.c -- CPU restatement of the intra path of the encode pass.  TEST INFRASTRUCTURE ONLY: nothing in the product links,
 * imports or calls this file (tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg are its only users).
 *
 * Follows, for blocks of 4x4 .. 32x32 that lie inside the picture (what encode_pass_sb codes when the picture's width and height
 * are multiples of 8; the 64x64 block is outside this restatement):
 *   reference samples   generate_intra_reference_samples     Source/Lib/Codec/EbEncDecProcess.c:1128-1310
 *                        (neighbour arrays written by eb_vp9_neighbor_array_unit_sample_write, Codec/EbNeighborArrays.c:107-240,
 *                         from the block's UNFILTERED reconstruction, EbEncDecProcess.c:4110-4160)
 *   predictors           Source/Lib/VPX/intrapred.c:22-416 (generic sizes and the 4x4 specials), chosen by
 *                        build_intra_predictors, Source/Lib/VPX/vp9_reconintra.c:249-408
 *   transform type       eb_vp9_intra_mode_to_tx_type_lookup, vp9_reconintra.c:20-31; get_tx_type / perform_coding_loop,
 *                        Codec/EbEncDecProcess.c:365-390 (chroma and 32x32: DCT_DCT)
 *   block order          encode_pass_sb, EbEncDecProcess.c:3680-4241: SBs in raster order, blocks in z-order, per block the three
 *                        predictions first, then luma / Cb / Cr transform + reconstruction, then the neighbour arrays
 * Pinned against the reference's own functions by tests/test_intra_oracle.py (oracle/_ref/ref_intra: generate_intra_reference_samples,
 * intra_prediction, perform_coding_loop and the neighbour-array writer, linked from the reference's files as they lie).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

#define AVG3(a, b, c) (((a) + 2 * (b) + (c) + 2) >> 2)
#define AVG2(a, b) (((a) + (b) + 1) >> 1)

/* VPX/intrapred.c:22-41 */
static void p_d207(uint8_t *dst, int stride, int bs, const uint8_t *above, const uint8_t *left) {
    (void)above;
    for (int r = 0; r < bs - 1; r++) dst[r * stride] = (uint8_t)AVG2(left[r], left[r + 1]);
    dst[(bs - 1) * stride] = left[bs - 1];
    dst++;
    for (int r = 0; r < bs - 2; r++) dst[r * stride] = (uint8_t)AVG3(left[r], left[r + 1], left[r + 2]);
    dst[(bs - 2) * stride] = (uint8_t)AVG3(left[bs - 2], left[bs - 1], left[bs - 1]);
    dst[(bs - 1) * stride] = left[bs - 1];
    dst++;
    for (int c = 0; c < bs - 2; c++) dst[(bs - 1) * stride + c] = left[bs - 1];
    for (int r = bs - 2; r >= 0; r--)
        for (int c = 0; c < bs - 2; c++) dst[r * stride + c] = dst[(r + 1) * stride + c - 2];
}
/* :43-57 */
static void p_d63(uint8_t *dst, int stride, int bs, const uint8_t *above, const uint8_t *left) {
    (void)left;
    for (int c = 0; c < bs; c++) {
        dst[c] = (uint8_t)AVG2(above[c], above[c + 1]);
        dst[stride + c] = (uint8_t)AVG3(above[c], above[c + 1], above[c + 2]);
    }
    for (int r = 2, size = bs - 2; r < bs; r += 2, --size) {
        memcpy(dst + (r + 0) * stride, dst + (r >> 1), (size_t)size);
        memset(dst + (r + 0) * stride + size, above[bs - 1], (size_t)(bs - size));
        memcpy(dst + (r + 1) * stride, dst + stride + (r >> 1), (size_t)size);
        memset(dst + (r + 1) * stride + size, above[bs - 1], (size_t)(bs - size));
    }
}
/* :59-73 */
static void p_d45(uint8_t *dst, int stride, int bs, const uint8_t *above, const uint8_t *left) {
    const uint8_t  above_right = above[bs - 1];
    const uint8_t *row0 = dst;
    (void)left;
    for (int x = 0; x < bs - 1; x++) dst[x] = (uint8_t)AVG3(above[x], above[x + 1], above[x + 2]);
    dst[bs - 1] = above_right;
    dst += stride;
    for (int x = 1, size = bs - 2; x < bs; x++, size--) {
        memcpy(dst, row0 + x, (size_t)size);
        memset(dst + size, above_right, (size_t)(x + 1));
        dst += stride;
    }
}
/* :75-97 */
static void p_d117(uint8_t *dst, int stride, int bs, const uint8_t *above, const uint8_t *left) {
    for (int c = 0; c < bs; c++) dst[c] = (uint8_t)AVG2(above[c - 1], above[c]);
    dst += stride;
    dst[0] = (uint8_t)AVG3(left[0], above[-1], above[0]);
    for (int c = 1; c < bs; c++) dst[c] = (uint8_t)AVG3(above[c - 2], above[c - 1], above[c]);
    dst += stride;
    dst[0] = (uint8_t)AVG3(above[-1], left[0], left[1]);
    for (int r = 3; r < bs; r++) dst[(r - 2) * stride] = (uint8_t)AVG3(left[r - 3], left[r - 2], left[r - 1]);
    for (int r = 2; r < bs; r++) {
        for (int c = 1; c < bs; c++) dst[c] = dst[-2 * stride + c - 1];
        dst += stride;
    }
}
/* :99-118 */
static void p_d135(uint8_t *dst, int stride, int bs, const uint8_t *above, const uint8_t *left) {
    uint8_t border[32 + 32 - 1];
    for (int i = 0; i < bs - 2; i++) border[i] = (uint8_t)AVG3(left[bs - 3 - i], left[bs - 2 - i], left[bs - 1 - i]);
    border[bs - 2] = (uint8_t)AVG3(above[-1], left[0], left[1]);
    border[bs - 1] = (uint8_t)AVG3(left[0], above[-1], above[0]);
    border[bs - 0] = (uint8_t)AVG3(above[-1], above[0], above[1]);
    for (int i = 0; i < bs - 2; i++) border[bs + 1 + i] = (uint8_t)AVG3(above[i], above[i + 1], above[i + 2]);
    for (int i = 0; i < bs; i++) memcpy(dst + i * stride, border + bs - 1 - i, (size_t)bs);
}
/* :120-138 */
static void p_d153(uint8_t *dst, int stride, int bs, const uint8_t *above, const uint8_t *left) {
    dst[0] = (uint8_t)AVG2(above[-1], left[0]);
    for (int r = 1; r < bs; r++) dst[r * stride] = (uint8_t)AVG2(left[r - 1], left[r]);
    dst++;
    dst[0] = (uint8_t)AVG3(left[0], above[-1], above[0]);
    dst[stride] = (uint8_t)AVG3(above[-1], left[0], left[1]);
    for (int r = 2; r < bs; r++) dst[r * stride] = (uint8_t)AVG3(left[r - 2], left[r - 1], left[r]);
    dst++;
    for (int c = 0; c < bs - 2; c++) dst[c] = (uint8_t)AVG3(above[c - 1], above[c], above[c + 1]);
    dst += stride;
    for (int r = 1; r < bs; r++) {
        for (int c = 0; c < bs - 2; c++) dst[c] = dst[-stride + c - 2];
        dst += stride;
    }
}
/* :140-172 */
static void p_v(uint8_t *dst, int stride, int bs, const uint8_t *above) {
    for (int r = 0; r < bs; r++) memcpy(dst + r * stride, above, (size_t)bs);
}
static void p_h(uint8_t *dst, int stride, int bs, const uint8_t *left) {
    for (int r = 0; r < bs; r++) memset(dst + r * stride, left[r], (size_t)bs);
}
static void p_tm(uint8_t *dst, int stride, int bs, const uint8_t *above, const uint8_t *left) {
    const int tl = above[-1];
    for (int r = 0; r < bs; r++)
        for (int c = 0; c < bs; c++) {
            const int v = left[r] + above[c] - tl;
            dst[r * stride + c] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
        }
}
/* :174-236: the four DC forms */
static void p_dc(uint8_t *dst, int stride, int bs, const uint8_t *above, const uint8_t *left, int have_left, int have_top) {
    int dc = 128;
    if (have_left || have_top) {
        int sum = 0, count = 0;
        if (have_top) { for (int i = 0; i < bs; i++) sum += above[i]; count += bs; }
        if (have_left) { for (int i = 0; i < bs; i++) sum += left[i]; count += bs; }
        dc = (sum + (count >> 1)) / count;
    }
    for (int r = 0; r < bs; r++) memset(dst + r * stride, dc, (size_t)bs);
}
/* the 4x4 forms the dispatch table names for TX_4X4 (vp9_reconintra.c:64-76 -> intrapred.c:270-416): they read the above-right
 * samples the generic forms do not */
#define D(x, y) dst[(x) + (y) * stride]
static void p_d63_4(uint8_t *dst, int stride, const uint8_t *a) {
    const int A = a[0], B = a[1], Cc = a[2], Dd = a[3], E = a[4], F = a[5], G = a[6];
    D(0, 0) = (uint8_t)AVG2(A, B);
    D(1, 0) = D(0, 2) = (uint8_t)AVG2(B, Cc);
    D(2, 0) = D(1, 2) = (uint8_t)AVG2(Cc, Dd);
    D(3, 0) = D(2, 2) = (uint8_t)AVG2(Dd, E);
    D(3, 2) = (uint8_t)AVG2(E, F);
    D(0, 1) = (uint8_t)AVG3(A, B, Cc);
    D(1, 1) = D(0, 3) = (uint8_t)AVG3(B, Cc, Dd);
    D(2, 1) = D(1, 3) = (uint8_t)AVG3(Cc, Dd, E);
    D(3, 1) = D(2, 3) = (uint8_t)AVG3(Dd, E, F);
    D(3, 3) = (uint8_t)AVG3(E, F, G);
}
static void p_d45_4(uint8_t *dst, int stride, const uint8_t *a) {
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) D(c, r) = (uint8_t)AVG3(a[r + c], a[r + c + 1], a[r + c + 2]);
    D(3, 3) = a[7];
}
#undef D

/* mode: PREDICTION_MODE (VPX/vp9_blockd.h): 0 DC, 1 V, 2 H, 3 D45, 4 D135, 5 D117, 6 D153, 7 D207, 8 D63, 9 TM.
 * above points at above_row[0] (above[-1] is the corner, 2 bs samples follow), left at left_col[0]. */
void svt_oracle_intra_predict(int32_t mode, int32_t bs, int32_t have_left, int32_t have_top, const uint8_t *above, const uint8_t *left,
                              uint8_t *dst, int32_t stride) {
    switch (mode) {
    case 0: p_dc(dst, stride, bs, above, left, have_left, have_top); break;
    case 1: p_v(dst, stride, bs, above); break;
    case 2: p_h(dst, stride, bs, left); break;
    case 3: if (bs == 4) p_d45_4(dst, stride, above); else p_d45(dst, stride, bs, above, left); break;
    case 4: p_d135(dst, stride, bs, above, left); break;
    case 5: p_d117(dst, stride, bs, above, left); break;
    case 6: p_d153(dst, stride, bs, above, left); break;
    case 7: p_d207(dst, stride, bs, above, left); break;
    case 8: if (bs == 4) p_d63_4(dst, stride, above); else p_d63(dst, stride, bs, above, left); break;
    default: p_tm(dst, stride, bs, above, left); break;
    }
}

/* generate_intra_reference_samples for a transform block of bs x bs at (x0, y0) of a plane whose blocks never use the above-right
 * neighbour (blocks >= 8x8: have_right = 0, EbEncDecProcess.c:1146) and never cross the picture edge (the "faster" paths).
 * above_row[0] = the corner, above_row[1 .. 2 bs] = the row; left_col[0 .. bs - 1]. */
void svt_oracle_intra_ref_samples2(const uint8_t *plane, int32_t stride, int32_t x0, int32_t y0, int32_t bs, int32_t have_right, uint8_t *above_row,
                                   uint8_t *left_col);
void svt_oracle_intra_ref_samples(const uint8_t *plane, int32_t stride, int32_t x0, int32_t y0, int32_t bs, uint8_t *above_row, uint8_t *left_col) {
    svt_oracle_intra_ref_samples2(plane, stride, x0, y0, bs, 0, above_row, left_col);
}
/* have_right: a 4x4 luma block in the left half of its 8x8 unit (aoff = 0: have_right = (aoff + txw) < bw, :1146) reads the four true
 * above-right samples (:1283-1288 and the const_above_row path :1279-1280, which is the same eight samples) */
void svt_oracle_intra_ref_samples2(const uint8_t *plane, int32_t stride, int32_t x0, int32_t y0, int32_t bs, int32_t have_right, uint8_t *above_row,
                                   uint8_t *left_col) {
    const int have_top = y0 > 0, have_left = x0 > 0;
    const uint8_t *p = plane + (size_t)y0 * stride + x0;
    if (have_left) for (int i = 0; i < bs; i++) left_col[i] = p[(ptrdiff_t)i * stride - 1];
    else memset(left_col, 129, (size_t)bs);
    if (have_top) {
        memcpy(above_row + 1, p - stride, (size_t)bs);
        if (bs == 4 && have_right) memcpy(above_row + 1 + bs, p - stride + bs, (size_t)bs);
        else memset(above_row + 1 + bs, above_row[bs], (size_t)bs);
        above_row[0] = have_left ? p[-stride - 1] : 129;
    } else {
        memset(above_row, 127, (size_t)(2 * bs + 1));
    }
}

static const uint8_t MODE_TX_TYPE[10] = {0, 1, 2, 0, 3, 1, 2, 2, 1, 3}; /* vp9_reconintra.c:20-31 (DCT_DCT 0, ADST_DCT 1, DCT_ADST 2, ADST_ADST 3) */

static uint32_t zorder4(int x4, int y4) {
    uint32_t v = 0;
    for (int b = 0; b < 4; b++) v |= (uint32_t)((x4 >> b) & 1) << (2 * b) | (uint32_t)((y4 >> b) & 1) << (2 * b + 1);
    return v;
}

/* One intra picture through the encode pass.  src / pred: the three tight planes one after the other (Y, U, V); the reconstruction
 * lives in recon_buf at recon_off[plane] with recon_stride[plane != 0].  mi: one record per 8x8 unit (sb_type 3 / 6 / 9, tx_size =
 * the block's own size, is_inter 0, pad_[1] = luma mode, pad_[2] = chroma mode; sb_type 0 = four 4x4 luma blocks + one 4x4 chroma
 * block per plane: tx_size 0, luma modes of blocks 0..3 in the nibbles of pad_[1] (0, 1) and pad_[0] (2, 3)).  qcoeff / dqcoeff: the product's position-addressed
 * layout (6144 per SB: luma 4096, Cb 1024, Cr 1024, 4x4 units in z-order); eob_map: one entry per 4x4 unit, Y then U then V planes,
 * written at the block's first unit.  iscan_off[tx_size * 4 + tx_type].  mixed: the picture is an inter picture whose inter blocks have
 * been reconstructed into recon_buf already (any block shape); only its intra blocks are coded, in the same order -- every intra block
 * then sees what the reference's neighbour arrays hold: the unfiltered reconstruction of the blocks coded before it, inter or intra
 * (encode_pass_sb writes them after every block, EbEncDecProcess.c:4110-4160).  Returns 0, or -1 for a grid outside this restatement. */
int32_t svt_oracle_intra_picture(const uint8_t *src, uint8_t *pred, uint8_t *recon_buf, const uint32_t recon_off[3], const int32_t recon_stride[2],
                                 const svt_lf_mode_info *mi, int32_t mi_stride, int32_t width, int32_t height, const svt_quant_tables qt[2],
                                 const int16_t *iscan, const uint32_t iscan_off[16], int16_t *qcoeff, int16_t *dqcoeff, uint16_t *eob_map, int32_t mixed) {
    const int mi_rows = height >> 3, mi_cols = width >> 3, sb_cols = (width + 63) >> 6, sb_rows = (height + 63) >> 6;
    const size_t po[3] = {0, (size_t)width * height, (size_t)width * height + (size_t)(width / 2) * (height / 2)};
    const size_t eo[3] = {0, (size_t)(width / 4) * (height / 4), (size_t)(width / 4) * (height / 4) + (size_t)(width / 8) * (height / 8)};
    for (int sr = 0; sr < sb_rows; sr++)
        for (int sc = 0; sc < sb_cols; sc++)
            for (int z = 0; z < 64; z++) {
                int r = 0, c = 0;
                for (int b = 0; b < 3; b++) { c |= ((z >> (2 * b)) & 1) << b; r |= ((z >> (2 * b + 1)) & 1) << b; }
                const int ur = sr * 8 + r, uc = sc * 8 + c;
                if (ur >= mi_rows || uc >= mi_cols) continue;
                const svt_lf_mode_info *b = &mi[ur * mi_stride + uc];
                if (b->is_inter && mixed) continue; /* an inter picture with intra blocks: the inter blocks are reconstructed already */
                if (b->is_inter || (b->sb_type != 0 && b->sb_type != 3 && b->sb_type != 6 && b->sb_type != 9)) return -1;
                const int w8 = b->sb_type == 0 ? 1 : 1 << ((b->sb_type - 3) / 3);
                if ((ur % w8) || (uc % w8)) continue;
                if (ur + w8 > mi_rows || uc + w8 > mi_cols || b->pad_[2] > 9) return -1;
                if (b->sb_type == 0 ? b->tx_size != 0 : (b->tx_size != (b->sb_type - 3) / 3 + 1 || b->pad_[1] > 9)) return -1;
                /* transform blocks of the prediction block in the reference's order: an 8x8 unit of 4x4 blocks is four blocks of its
                   own (bmi_index 0..3, :3706), the chroma 4x4 rides with the last; any other block is luma, Cb, Cr */
                const int sub = b->sb_type == 0;
                const int modes4[4] = {b->pad_[1] & 15, b->pad_[1] >> 4, b->pad_[0] & 15, b->pad_[0] >> 4}; /* 4x4: luma modes of blocks 0..3 */
                for (int step = 0; step < (sub ? 6 : 3); step++) {
                    const int plane = sub ? (step < 4 ? 0 : step - 3) : step;
                    const int bmi = sub && step < 4 ? step : 0;
                    const int bs = sub ? 4 : (plane ? w8 * 4 : w8 * 8);
                    const int x0 = (plane ? uc * 4 : uc * 8) + (sub && !plane ? 4 * (bmi & 1) : 0), y0 = (plane ? ur * 4 : ur * 8) + (sub && !plane ? 4 * (bmi >> 1) : 0);
                    const int pw = plane ? width / 2 : width, rs = recon_stride[plane ? 1 : 0];
                    const int mode = plane ? b->pad_[2] : sub ? modes4[bmi] : b->pad_[1];
                    if (mode > 9) return -1;
                    uint8_t  *rp = recon_buf + recon_off[plane];
                    uint8_t   above[65], left[32];
                    svt_oracle_intra_ref_samples2(rp, rs, x0, y0, bs, sub && !plane && !(bmi & 1), above, left);
                    svt_oracle_intra_predict(mode, bs, x0 > 0, y0 > 0, above + 1, left, pred + po[plane] + (size_t)y0 * pw + x0, pw);
                    svt_tq_block k;
                    memset(&k, 0, sizeof k);
                    k.src_off = k.pred_off = (uint32_t)(po[plane] + (size_t)y0 * pw + x0);
                    k.recon_off = recon_off[plane] + (uint32_t)y0 * (uint32_t)rs + (uint32_t)x0;
                    k.src_stride = k.pred_stride = (uint16_t)pw; k.recon_stride = (uint16_t)rs;
                    k.tx_size = (uint8_t)(sub ? 0 : plane ? b->tx_size - 1 : b->tx_size);
                    k.tx_type = (uint8_t)((plane == 0 && k.tx_size < 3) ? MODE_TX_TYPE[mode] : 0);
                    k.qtab = (uint8_t)(plane ? 1 : 0); k.do_recon = 1;
                    k.iscan_off = iscan_off[k.tx_size * 4 + k.tx_type];
                    const int sbw = plane ? 32 : 64;
                    k.coeff_off = (uint32_t)(sr * sb_cols + sc) * 6144u + (plane == 0 ? 0u : plane == 1 ? 4096u : 5120u) + zorder4((x0 % sbw) >> 2, (y0 % sbw) >> 2) * 16u;
                    uint16_t eob = 0;
                    if (svt_oracle_tq_batch(src, pred, recon_buf, &k, 1, qt, iscan, qcoeff, dqcoeff, &eob)) return -1;
                    eob_map[eo[plane] + (size_t)(y0 >> 2) * (size_t)(pw >> 2) + (size_t)(x0 >> 2)] = eob;
                }
            }
    return 0;
}
