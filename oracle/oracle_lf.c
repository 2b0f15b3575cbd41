/*
 * oracle_lf.c -- CPU restatement of SVT-VP9's in-loop deblocking filter (C path).
 *
 * TEST INFRASTRUCTURE ONLY (see oracle_me.c).  Restates, relative to /root/reference/Source/Lib/VPX:
 *   edge arithmetic      loopfilter.c:31-327 (filter_mask, flat_mask4/5, hev_mask, filter4/8/16 and the
 *                        eb_vp9_lpf_{horizontal,vertical}_{4,8,16}[_dual]_c wrappers)
 *   per-SB edge walk     vp9_loopfilter.c:305-397 filter_selectively_vert_row2, :481-568 filter_selectively_horiz,
 *                        :1238-1341 eb_vp9_filter_block_plane_ss00, :1342-1455 _ss11
 *   frame driver         vp9_loopfilter.c:786-900 eb_vp9_adjust_mask, :1456-1546 loop_filter_rows / eb_vp9_loop_filter_frame
 *   thresholds           vp9_loopfilter.c:221-262 update_sharpness / eb_vp9_loop_filter_init
 *   level from q         vp9_picklpf.c:37-89 eb_vp9_pick_filter_level (LPF_PICK_FROM_Q, 8-bit)
 * Pinned against the reference: edge filters through oracle/_ref/libsvtref_kernels.so, the whole frame through
 * oracle/_ref/ref_lf_frame (tests/test_oracle_vs_ref.py).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/svtvp9_hip.h"
#include "oracle.h"

static inline int iabs(int v) { return v < 0 ? -v : v; }
static inline int8_t sclamp(int t) { return (int8_t)(t < -128 ? -128 : t > 127 ? 127 : t); }

/* one pixel position across an edge: p[-k*st] = p(k-1).., p[k*st] = q(k) ; kind 4 / 8 / 16 */
static void filter_px(uint8_t *s, int st, int kind, int mblim, int lim, int hev_thr) {
    const int p3 = s[-4 * st], p2 = s[-3 * st], p1 = s[-2 * st], p0 = s[-st];
    const int q0 = s[0], q1 = s[st], q2 = s[2 * st], q3 = s[3 * st];
    /* filter_mask */
    int mask = !(iabs(p3 - p2) > lim || iabs(p2 - p1) > lim || iabs(p1 - p0) > lim || iabs(q1 - q0) > lim || iabs(q2 - q1) > lim ||
                 iabs(q3 - q2) > lim || iabs(p0 - q0) * 2 + iabs(p1 - q1) / 2 > mblim);
    int flat = 0, flat2 = 0;
    if (kind >= 8)
        flat = !(iabs(p1 - p0) > 1 || iabs(q1 - q0) > 1 || iabs(p2 - p0) > 1 || iabs(q2 - q0) > 1 || iabs(p3 - p0) > 1 || iabs(q3 - q0) > 1);
    if (kind == 16) {
        const int p7 = s[-8 * st], p6 = s[-7 * st], p5 = s[-6 * st], p4 = s[-5 * st];
        const int q4 = s[4 * st], q5 = s[5 * st], q6 = s[6 * st], q7 = s[7 * st];
        /* flat_mask5(1, p7, p6, p5, p4, p0, q0, q4, q5, q6, q7) */
        flat2 = !(iabs(p4 - p0) > 1 || iabs(q4 - q0) > 1 || iabs(p5 - p0) > 1 || iabs(q5 - q0) > 1 || iabs(p6 - p0) > 1 || iabs(q6 - q0) > 1 ||
                  iabs(p7 - p0) > 1 || iabs(q7 - q0) > 1);
        if (flat2 && flat && mask) {
#define R4(x) (uint8_t)(((x) + 8) >> 4)
            s[-7 * st] = R4(p7 * 7 + p6 * 2 + p5 + p4 + p3 + p2 + p1 + p0 + q0);
            s[-6 * st] = R4(p7 * 6 + p6 + p5 * 2 + p4 + p3 + p2 + p1 + p0 + q0 + q1);
            s[-5 * st] = R4(p7 * 5 + p6 + p5 + p4 * 2 + p3 + p2 + p1 + p0 + q0 + q1 + q2);
            s[-4 * st] = R4(p7 * 4 + p6 + p5 + p4 + p3 * 2 + p2 + p1 + p0 + q0 + q1 + q2 + q3);
            s[-3 * st] = R4(p7 * 3 + p6 + p5 + p4 + p3 + p2 * 2 + p1 + p0 + q0 + q1 + q2 + q3 + q4);
            s[-2 * st] = R4(p7 * 2 + p6 + p5 + p4 + p3 + p2 + p1 * 2 + p0 + q0 + q1 + q2 + q3 + q4 + q5);
            s[-1 * st] = R4(p7 + p6 + p5 + p4 + p3 + p2 + p1 + p0 * 2 + q0 + q1 + q2 + q3 + q4 + q5 + q6);
            s[0 * st]  = R4(p6 + p5 + p4 + p3 + p2 + p1 + p0 + q0 * 2 + q1 + q2 + q3 + q4 + q5 + q6 + q7);
            s[1 * st]  = R4(p5 + p4 + p3 + p2 + p1 + p0 + q0 + q1 * 2 + q2 + q3 + q4 + q5 + q6 + q7 * 2);
            s[2 * st]  = R4(p4 + p3 + p2 + p1 + p0 + q0 + q1 + q2 * 2 + q3 + q4 + q5 + q6 + q7 * 3);
            s[3 * st]  = R4(p3 + p2 + p1 + p0 + q0 + q1 + q2 + q3 * 2 + q4 + q5 + q6 + q7 * 4);
            s[4 * st]  = R4(p2 + p1 + p0 + q0 + q1 + q2 + q3 + q4 * 2 + q5 + q6 + q7 * 5);
            s[5 * st]  = R4(p1 + p0 + q0 + q1 + q2 + q3 + q4 + q5 * 2 + q6 + q7 * 6);
            s[6 * st]  = R4(p0 + q0 + q1 + q2 + q3 + q4 + q5 + q6 * 2 + q7 * 7);
#undef R4
            return;
        }
    }
    if (flat && mask) {
#define R3(x) (uint8_t)(((x) + 4) >> 3)
        s[-3 * st] = R3(p3 + p3 + p3 + 2 * p2 + p1 + p0 + q0);
        s[-2 * st] = R3(p3 + p3 + p2 + 2 * p1 + p0 + q0 + q1);
        s[-1 * st] = R3(p3 + p2 + p1 + 2 * p0 + q0 + q1 + q2);
        s[0 * st]  = R3(p2 + p1 + p0 + 2 * q0 + q1 + q2 + q3);
        s[1 * st]  = R3(p1 + p0 + q0 + 2 * q1 + q2 + q3 + q3);
        s[2 * st]  = R3(p0 + q0 + q1 + 2 * q2 + q3 + q3 + q3);
#undef R3
        return;
    }
    /* filter4 (int8 arithmetic, mask/hev as 0 / -1) */
    const int8_t m = mask ? -1 : 0;
    const int8_t hev = (iabs(p1 - p0) > hev_thr || iabs(q1 - q0) > hev_thr) ? -1 : 0;
    const int8_t ps1 = (int8_t)(p1 ^ 0x80), ps0 = (int8_t)(p0 ^ 0x80), qs0 = (int8_t)(q0 ^ 0x80), qs1 = (int8_t)(q1 ^ 0x80);
    int8_t f = (int8_t)(sclamp(ps1 - qs1) & hev);
    f = (int8_t)(sclamp(f + 3 * (qs0 - ps0)) & m);
    const int8_t f1 = (int8_t)(sclamp(f + 4) >> 3), f2 = (int8_t)(sclamp(f + 3) >> 3);
    s[0]   = (uint8_t)(sclamp(qs0 - f1) ^ 0x80);
    s[-st] = (uint8_t)(sclamp(ps0 + f2) ^ 0x80);
    f = (int8_t)(((f1 + 1) >> 1) & ~hev);
    s[st]      = (uint8_t)(sclamp(qs1 - f) ^ 0x80);
    s[-2 * st] = (uint8_t)(sclamp(ps1 + f) ^ 0x80);
}

/* `count` pixels along an edge; across = step across the edge, along = step along it */
void oracle_lpf_edge(uint8_t *s, int across, int along, int count, int kind, int mblim, int lim, int hev_thr) {
    for (int i = 0; i < count; i++) filter_px(s + i * along, across, kind, mblim, lim, hev_thr);
}

void svt_oracle_lf_thresh_init(svt_lf_thresh *t, int32_t sharp) {
    for (int lvl = 0; lvl <= 63; lvl++) {
        int lim = lvl >> ((sharp > 0) + (sharp > 4));
        if (sharp > 0 && lim > 9 - sharp) lim = 9 - sharp;
        if (lim < 1) lim = 1;
        t->lim[lvl]     = (uint8_t)lim;
        t->mblim[lvl]   = (uint8_t)(2 * (lvl + 2) + lim);
        t->hev_thr[lvl] = (uint8_t)(lvl >> 4);
    }
}
int32_t svt_oracle_lf_level_from_q(int32_t q, int32_t is_key) {
    int g = (q * 20723 + 1015158 + (1 << 17)) >> 18;
    if (is_key) g -= 4;
    return g < 0 ? 0 : g > 63 ? 63 : g;
}

/* eb_vp9_adjust_mask, vp9_loopfilter.c:786-900 */
static void adjust_mask(svt_lf_mask *m, int mi_row, int mi_col, int mi_rows, int mi_cols) {
    m->left_y[2] |= m->left_y[3]; m->above_y[2] |= m->above_y[3];
    m->left_uv[2] |= m->left_uv[3]; m->above_uv[2] |= m->above_uv[3];
    m->left_y[1] |= m->left_y[0] & 0x1111111111111111ULL; m->left_y[0] &= ~0x1111111111111111ULL;
    m->above_y[1] |= m->above_y[0] & 0x000000ff000000ffULL; m->above_y[0] &= ~0x000000ff000000ffULL;
    m->left_uv[1] |= m->left_uv[0] & 0x1111; m->left_uv[0] &= (uint16_t)~0x1111;
    m->above_uv[1] |= m->above_uv[0] & 0x000f; m->above_uv[0] &= (uint16_t)~0x000f;
    if (mi_row + 8 > mi_rows) {
        const uint64_t rows = (uint64_t)(mi_rows - mi_row);
        const uint64_t my = (((uint64_t)1 << (rows << 3)) - 1);
        const uint16_t muv = (uint16_t)(((uint16_t)1 << (((rows + 1) >> 1) << 2)) - 1);
        for (int i = 0; i < 3; i++) { m->left_y[i] &= my; m->above_y[i] &= my; m->left_uv[i] &= muv; m->above_uv[i] &= muv; }
        m->int_4x4_y &= my; m->int_4x4_uv &= muv;
        if (rows == 1) { m->above_uv[1] |= m->above_uv[2]; m->above_uv[2] = 0; }
        if (rows == 5) { m->above_uv[1] |= m->above_uv[2] & 0xff00; m->above_uv[2] &= (uint16_t)~(m->above_uv[2] & 0xff00); }
    }
    if (mi_col + 8 > mi_cols) {
        const uint64_t cols = (uint64_t)(mi_cols - mi_col);
        const uint64_t my = (uint64_t)((1 << cols) - 1) * 0x0101010101010101ULL;
        const uint16_t muv = (uint16_t)(((1 << ((cols + 1) >> 1)) - 1) * 0x1111);
        const uint16_t muvi = (uint16_t)(((1 << (cols >> 1)) - 1) * 0x1111);
        for (int i = 0; i < 3; i++) { m->left_y[i] &= my; m->above_y[i] &= my; m->left_uv[i] &= muv; m->above_uv[i] &= muv; }
        m->int_4x4_y &= my; m->int_4x4_uv &= muvi;
        if (cols == 1) { m->left_uv[1] |= m->left_uv[2]; m->left_uv[2] = 0; }
        if (cols == 5) { m->left_uv[1] |= (m->left_uv[2] & 0xcccc); m->left_uv[2] &= (uint16_t)~(m->left_uv[2] & 0xcccc); }
    }
    if (mi_col == 0)
        for (int i = 0; i < 3; i++) { m->left_y[i] &= 0xfefefefefefefefeULL; m->left_uv[i] &= 0xeeee; }
}

/* filter_selectively_vert_row2: two 8-pixel rows, all column positions left to right.
 * [quirk] a 16-wide edge set in both rows is filtered over 16 rows with the FIRST row's thresholds
 * (eb_vpx_lpf_vertical_16_dual takes one threshold set, vp9_loopfilter.c:325-327). */
/* rows_avail: picture rows that exist below s (the reference filters 8/16 rows regardless and relies on the
 * recon buffer's padding when a chroma SB row is only 4 rows high; rows are independent in this pass, so
 * stopping at the picture edge leaves every in-picture sample identical) */
static void vert_row2(int ss, uint8_t *s, int pitch, unsigned m16, unsigned m8, unsigned m4, unsigned mi4, const svt_lf_thresh *t,
                      const uint8_t *lfl, int rows_avail) {
    const int fwd = ss ? 4 : 8;
    const unsigned cutoff = ss ? 0xff : 0xffff, one = 1u | (1u << fwd);
    for (unsigned mask = (m16 | m8 | m4 | mi4) & cutoff; mask; mask = (mask & ~one) >> 1) {
        if (mask & one) {
            const int l[2] = {lfl[0], lfl[fwd]};
            uint8_t  *r[2] = {s, s + 8 * pitch};
            int       n[2] = {rows_avail < 8 ? rows_avail : 8, rows_avail - 8 < 0 ? 0 : rows_avail - 8 < 8 ? rows_avail - 8 : 8};
            if (m16 & one) {
                if ((m16 & one) == one) oracle_lpf_edge(r[0], 1, pitch, n[0] + n[1], 16, t->mblim[l[0]], t->lim[l[0]], t->hev_thr[l[0]]);
                else { int h = !(m16 & 1); oracle_lpf_edge(r[h], 1, pitch, n[h], 16, t->mblim[l[h]], t->lim[l[h]], t->hev_thr[l[h]]); }
            }
            for (int h = 0; h < 2; h++) {
                const unsigned bit = h ? (1u << fwd) : 1u;
                if (m8 & bit) oracle_lpf_edge(r[h], 1, pitch, n[h], 8, t->mblim[l[h]], t->lim[l[h]], t->hev_thr[l[h]]);
            }
            for (int h = 0; h < 2; h++) {
                const unsigned bit = h ? (1u << fwd) : 1u;
                if (m4 & bit) oracle_lpf_edge(r[h], 1, pitch, n[h], 4, t->mblim[l[h]], t->lim[l[h]], t->hev_thr[l[h]]);
            }
            for (int h = 0; h < 2; h++) {
                const unsigned bit = h ? (1u << fwd) : 1u;
                if (mi4 & bit) oracle_lpf_edge(r[h] + 4, 1, pitch, n[h], 4, t->mblim[l[h]], t->lim[l[h]], t->hev_thr[l[h]]);
            }
        }
        s += 8; lfl += 1; m16 >>= 1; m8 >>= 1; m4 >>= 1; mi4 >>= 1;
    }
}

/* filter_selectively_horiz: one 8-pixel row, column blocks left to right with greedy pairing.
 * [quirk] the second block of a 16-wide pair uses the first block's thresholds (vp9_loopfilter.c:492-494). */
#define HEDGE(ptr, colofs, kind_, l_) do { int n_ = cols_avail - (colofs); if (n_ > 8) n_ = 8; \
        if (n_ > 0) oracle_lpf_edge((ptr), pitch, 1, n_, (kind_), t->mblim[l_], t->lim[l_], t->hev_thr[l_]); } while (0)
/* cols_avail: picture columns that exist right of s (same argument as rows_avail, for a 4-column chroma block) */
static void horiz_row(uint8_t *s, int pitch, unsigned m16, unsigned m8, unsigned m4, unsigned mi4, const svt_lf_thresh *t, const uint8_t *lfl,
                      int cols_avail) {
    int count;
    for (unsigned mask = m16 | m8 | m4 | mi4; mask; mask >>= count) {
        count = 1;
        if (mask & 1) {
            const int l0 = lfl[0];
            if (m16 & 1) {
                if ((m16 & 3) == 3) { HEDGE(s, 0, 16, l0); HEDGE(s + 8, 8, 16, l0); count = 2; }
                else HEDGE(s, 0, 16, l0);
            } else if ((m8 & 1) || (m4 & 1)) {
                const int      kind = (m8 & 1) ? 8 : 4;
                const unsigned mk = (m8 & 1) ? m8 : m4;
                if ((mk & 3) == 3) {
                    const int l1 = lfl[1];
                    HEDGE(s, 0, kind, l0);
                    HEDGE(s + 8, 8, kind, l1);
                    if ((mi4 & 3) == 3) { HEDGE(s + 4 * pitch, 0, 4, l0); HEDGE(s + 8 + 4 * pitch, 8, 4, l1); }
                    else if (mi4 & 1) HEDGE(s + 4 * pitch, 0, 4, l0);
                    else if (mi4 & 2) HEDGE(s + 8 + 4 * pitch, 8, 4, l1);
                    count = 2;
                } else {
                    HEDGE(s, 0, kind, l0);
                    if (mi4 & 1) HEDGE(s + 4 * pitch, 0, 4, l0);
                }
            } else {
                HEDGE(s + 4 * pitch, 0, 4, l0);
            }
        }
        s += 8 * count; lfl += count; m16 >>= count; m8 >>= count; m4 >>= count; mi4 >>= count; cols_avail -= 8 * count;
    }
}

/* eb_vp9_filter_block_plane_ss00 / _ss11 */
static void block_plane_y(uint8_t *buf, int stride, int mi_row, int mi_rows, const svt_lf_mask *m, const svt_lf_thresh *t, int rows_avail, int cols_avail) {
    uint64_t m16 = m->left_y[2], m8 = m->left_y[1], m4 = m->left_y[0], mi = m->int_4x4_y;
    uint8_t *b = buf;
    for (int r = 0; r < 8 && mi_row + r < mi_rows; r += 2) {
        vert_row2(0, b, stride, (unsigned)m16, (unsigned)m8, (unsigned)m4, (unsigned)mi, t, &m->lfl_y[r << 3], rows_avail - 8 * r);
        b += 16 * stride; m16 >>= 16; m8 >>= 16; m4 >>= 16; mi >>= 16;
    }
    b = buf; m16 = m->above_y[2]; m8 = m->above_y[1]; m4 = m->above_y[0]; mi = m->int_4x4_y;
    for (int r = 0; r < 8 && mi_row + r < mi_rows; r++) {
        unsigned a16 = 0, a8 = 0, a4 = 0;
        if (mi_row + r != 0) { a16 = (unsigned)(m16 & 0xff); a8 = (unsigned)(m8 & 0xff); a4 = (unsigned)(m4 & 0xff); }
        horiz_row(b, stride, a16, a8, a4, (unsigned)(mi & 0xff), t, &m->lfl_y[r << 3], cols_avail);
        b += 8 * stride; m16 >>= 8; m8 >>= 8; m4 >>= 8; mi >>= 8;
    }
}
static void block_plane_uv(uint8_t *buf, int stride, int mi_row, int mi_rows, const svt_lf_mask *m, const svt_lf_thresh *t, int rows_avail, int cols_avail) {
    uint16_t m16 = m->left_uv[2], m8 = m->left_uv[1], m4 = m->left_uv[0], mi = m->int_4x4_uv;
    uint8_t  lfl_uv[16];
    uint8_t *b = buf;
    memset(lfl_uv, 0, sizeof lfl_uv);
    for (int r = 0; r < 8 && mi_row + r < mi_rows; r += 4) {
        for (int c = 0; c < 4; c++) {
            lfl_uv[(r << 1) + c]       = m->lfl_y[(r << 3) + (c << 1)];
            lfl_uv[((r + 2) << 1) + c] = m->lfl_y[((r + 2) << 3) + (c << 1)];
        }
        vert_row2(1, b, stride, m16, m8, m4, mi, t, &lfl_uv[r << 1], rows_avail - 4 * r);
        b += 16 * stride; m16 >>= 8; m8 >>= 8; m4 >>= 8; mi >>= 8;
    }
    b = buf; m16 = m->above_uv[2]; m8 = m->above_uv[1]; m4 = m->above_uv[0]; mi = m->int_4x4_uv;
    for (int r = 0; r < 8 && mi_row + r < mi_rows; r += 2) {
        const int      skip = mi_row + r == mi_rows - 1;
        const unsigned mi_r = skip ? 0 : (unsigned)(mi & 0xf);
        unsigned       a16 = 0, a8 = 0, a4 = 0;
        if (mi_row + r != 0) { a16 = m16 & 0xf; a8 = m8 & 0xf; a4 = m4 & 0xf; }
        horiz_row(b, stride, a16, a8, a4, mi_r, t, &lfl_uv[r << 1], cols_avail);
        b += 8 * stride; m16 >>= 4; m8 >>= 4; m4 >>= 4; mi >>= 4;
    }
}

/* eb_vp9_loop_filter_frame (partial_frame = 0): SB raster order; the masks are adjusted per SB on a copy */
int32_t svt_oracle_lf_frame(const svt_yuv_planes *recon, const svt_lf_mask *lfm, int32_t lfm_stride, const svt_lf_thresh *thr,
                            int32_t mi_rows, int32_t mi_cols, int32_t y_only) {
    for (int mi_row = 0; mi_row < mi_rows; mi_row += 8)
        for (int mi_col = 0; mi_col < mi_cols; mi_col += 8) {
            svt_lf_mask m = lfm[(mi_row >> 3) * lfm_stride + (mi_col >> 3)];
            adjust_mask(&m, mi_row, mi_col, mi_rows, mi_cols);
            block_plane_y(recon->y + (size_t)(mi_row * 8) * recon->y_stride + mi_col * 8, recon->y_stride, mi_row, mi_rows, &m, thr,
                          recon->height - mi_row * 8, recon->width - mi_col * 8);
            if (!y_only) {
                block_plane_uv(recon->u + (size_t)(mi_row * 4) * recon->uv_stride + mi_col * 4, recon->uv_stride, mi_row, mi_rows, &m, thr,
                               recon->height / 2 - mi_row * 4, recon->width / 2 - mi_col * 4);
                block_plane_uv(recon->v + (size_t)(mi_row * 4) * recon->uv_stride + mi_col * 4, recon->uv_stride, mi_row, mi_rows, &m, thr,
                               recon->height / 2 - mi_row * 4, recon->width / 2 - mi_col * 4);
            }
        }
    return 0;
}


/* ------------------------------------------------------------------------------------------------------------------ */
/* L2: LOOP_FILTER_MASK construction, frame-level formulation: eb_vp9_build_mask_frame (VPX/vp9_loopfilter.c:1548-1571) */
/* -> eb_vp9_setup_mask (:901-1040): walk the 64/32/16/8 partition tree of every SB, build_masks (:706-783) for blocks   */
/* that carry chroma edges, build_y_mask (:788-825) for the second/third/fourth block of a 16x16 area.                  */
/* ------------------------------------------------------------------------------------------------------------------ */
static const uint8_t lf_w8[13] = {1, 1, 1, 1, 1, 2, 2, 2, 4, 4, 4, 8, 8}; /* eb_vp9_num_8x8_blocks_wide_lookup */
static const uint8_t lf_h8[13] = {1, 1, 1, 1, 2, 1, 2, 4, 2, 4, 8, 4, 8}; /* eb_vp9_num_8x8_blocks_high_lookup */

static uint64_t lf_rect(int w, int h, int cols) {
    uint64_t m = 0;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) m |= (uint64_t)1 << (y * cols + x);
    return m;
}
/* left_64x64_txform_mask / above_64x64_txform_mask and their uv versions (:95-157) */
static uint64_t lf_txmask(int tx, int left, int cols) {
    uint64_t m = 0;
    const int step = tx < 2 ? 1 : tx == 2 ? 2 : 4;
    for (int y = 0; y < cols; y++)
        for (int x = 0; x < cols; x++)
            if ((left ? x : y) % step == 0) m |= (uint64_t)1 << (y * cols + x);
    return m;
}
/* eb_vp9_uv_txsize_lookup[bsize][tx][1][1] (VPX/vp9_common_data.c): min(tx, largest tx of the 4:2:0 chroma block) */
static int lf_uv_tx(int bs, int tx) {
    static const uint8_t max_uv[13] = {0, 0, 0, 0, 0, 0, 1, 1, 1, 2, 2, 2, 3};
    return tx < max_uv[bs] ? tx : max_uv[bs];
}

static void lf_build(const svt_lf_mode_info *b, int shift_y, int shift_uv, int with_uv, svt_lf_mask *m) {
    const int bs = b->sb_type, txy = b->tx_size, txuv = lf_uv_tx(bs, txy), w = lf_w8[bs], h = lf_h8[bs];
    const int wuv = (w + 1) >> 1, huv = (h + 1) >> 1;
    if (!b->filter_level) return;
    for (int i = 0; i < h; i++) memset(&m->lfl_y[shift_y + 8 * i], b->filter_level, (size_t)w);
    m->above_y[txy] |= lf_rect(w, 1, 8) << shift_y;
    m->left_y[txy] |= lf_rect(1, h, 8) << shift_y;
    if (with_uv) {
        m->above_uv[txuv] |= (uint16_t)(lf_rect(wuv, 1, 4) << shift_uv);
        m->left_uv[txuv] |= (uint16_t)(lf_rect(1, huv, 4) << shift_uv);
    }
    if (b->skip && b->is_inter) return;
    m->above_y[txy] |= (lf_rect(w, h, 8) & lf_txmask(txy, 0, 8)) << shift_y;
    m->left_y[txy] |= (lf_rect(w, h, 8) & lf_txmask(txy, 1, 8)) << shift_y;
    if (txy == 0) m->int_4x4_y |= lf_rect(w, h, 8) << shift_y;
    if (with_uv) {
        m->above_uv[txuv] |= (uint16_t)((lf_rect(wuv, huv, 4) & lf_txmask(txuv, 0, 4)) << shift_uv);
        m->left_uv[txuv] |= (uint16_t)((lf_rect(wuv, huv, 4) & lf_txmask(txuv, 1, 4)) << shift_uv);
        if (txuv == 0) m->int_4x4_uv |= (uint16_t)(lf_rect(wuv, huv, 4) << shift_uv);
    }
}

int32_t svt_oracle_lf_build_masks(const svt_lf_mode_info *mi, int32_t mi_stride, int32_t mi_rows, int32_t mi_cols,
                                  svt_lf_mask *lfm, int32_t lfm_stride) {
    enum { B64X64 = 12, B64X32 = 11, B32X64 = 10, B32X32 = 9, B32X16 = 8, B16X32 = 7, B16X16 = 6, B16X8 = 5, B8X16 = 4 };
    for (int mi_row = 0; mi_row < mi_rows; mi_row += 8)
        for (int mi_col = 0; mi_col < mi_cols; mi_col += 8) {
            svt_lf_mask            *m = &lfm[(mi_row >> 3) * lfm_stride + (mi_col >> 3)];
            const svt_lf_mode_info *p = mi + (size_t)mi_row * mi_stride + mi_col;
            const int max_rows = mi_rows - mi_row < 8 ? mi_rows - mi_row : 8, max_cols = mi_cols - mi_col < 8 ? mi_cols - mi_col : 8;
#define AT(r, c) (&p[(r) * mi_stride + (c)])
            memset(m, 0, sizeof *m);
            const int t64 = AT(0, 0)->sb_type;
            if (t64 == B64X64) { lf_build(AT(0, 0), 0, 0, 1, m); continue; }
            if (t64 == B64X32) { lf_build(AT(0, 0), 0, 0, 1, m); if (4 < max_rows) lf_build(AT(4, 0), 32, 8, 1, m); continue; }
            if (t64 == B32X64) { lf_build(AT(0, 0), 0, 0, 1, m); if (4 < max_cols) lf_build(AT(0, 4), 4, 2, 1, m); continue; }
            for (int q32 = 0; q32 < 4; q32++) {
                const int r32 = (q32 >> 1) * 4, c32 = (q32 & 1) * 4;
                if (c32 >= max_cols || r32 >= max_rows) continue;
                const int t32 = AT(r32, c32)->sb_type, sy32 = r32 * 8 + c32, suv32 = (r32 >> 1) * 4 + (c32 >> 1);
                if (t32 == B32X32) { lf_build(AT(r32, c32), sy32, suv32, 1, m); continue; }
                if (t32 == B32X16) { lf_build(AT(r32, c32), sy32, suv32, 1, m); if (r32 + 2 < max_rows) lf_build(AT(r32 + 2, c32), sy32 + 16, suv32 + 4, 1, m); continue; }
                if (t32 == B16X32) { lf_build(AT(r32, c32), sy32, suv32, 1, m); if (c32 + 2 < max_cols) lf_build(AT(r32, c32 + 2), sy32 + 2, suv32 + 1, 1, m); continue; }
                for (int q16 = 0; q16 < 4; q16++) {
                    const int r16 = r32 + (q16 >> 1) * 2, c16 = c32 + (q16 & 1) * 2;
                    if (c16 >= max_cols || r16 >= max_rows) continue;
                    const int t16 = AT(r16, c16)->sb_type, sy16 = r16 * 8 + c16, suv16 = (r16 >> 1) * 4 + (c16 >> 1);
                    if (t16 == B16X16) { lf_build(AT(r16, c16), sy16, suv16, 1, m); continue; }
                    if (t16 == B16X8) { lf_build(AT(r16, c16), sy16, suv16, 1, m); if (r16 + 1 < max_rows) lf_build(AT(r16 + 1, c16), sy16 + 8, 0, 0, m); continue; }
                    if (t16 == B8X16) { lf_build(AT(r16, c16), sy16, suv16, 1, m); if (c16 + 1 < max_cols) lf_build(AT(r16, c16 + 1), sy16 + 1, 0, 0, m); continue; }
                    lf_build(AT(r16, c16), sy16, suv16, 1, m);
                    for (int q8 = 1; q8 < 4; q8++) {
                        const int r8 = r16 + (q8 >> 1), c8 = c16 + (q8 & 1);
                        if (c8 >= max_cols || r8 >= max_rows) continue;
                        lf_build(AT(r8, c8), r8 * 8 + c8, 0, 0, m);
                    }
                }
            }
#undef AT
        }
    return 0;
}
