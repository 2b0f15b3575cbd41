/*
 * lf_kernel.hip -- VP9 in-loop deblocking of whole frames on gfx950.
 *
 * Replaces eb_vp9_loop_filter_frame (Source/Lib/VPX/vp9_loopfilter.c:1521 -> loop_filter_rows :1456 ->
 * eb_vp9_adjust_mask :786, eb_vp9_filter_block_plane_ss00 :1238 / _ss11 :1342 -> filter_selectively_vert_row2 :305,
 * filter_selectively_horiz :481 -> edge filters VPX/loopfilter.c:31-327).
 *
 * Ordering.  The reference filters super-blocks in raster order, all vertical edges of an SB, then all its
 * horizontal edges.  Bit-exactness only needs the true data dependencies: SB (r, c) must run after (r, c-1)
 * (its left-edge filters read/modify the neighbour's last columns, already filtered both ways) and after
 * (r-1, c+1) (whose left-edge filters modify the columns of (r-1, c) that (r, c)'s top-edge filters touch).
 * One workgroup owns one SB row and walks it left to right; rows form a wavefront (2 SBs lag per row) synchronised
 * through per-row progress counters in HBM with agent-scope release/acquire (MI355X: per-XCD L2s are not coherent).
 * Workgroups take their row in ticket order, so a row's predecessor has always started: no residency assumption.
 *
 * Inside an SB: the 72x72 luma tile (8-sample halo up/left) and the two 40x40 chroma tiles are staged in LDS;
 * wave 0 filters luma (lane = sample row for vertical edges, lane = sample column for horizontal edges), wave 1
 * filters Cb (lanes 0-31) and Cr (lanes 32-63).  Byte/integer VALU work, no MFMA.
 */
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "svt_ctx.h"

namespace {

struct lf_pic_dev {
    svt_yuv_planes     planes;
    const svt_lf_mask *lfm;
    int32_t            lfm_stride, mi_rows, mi_cols, y_only;
    uint32_t          *progress; /* [sb_rows] SBs completed per SB row */
};

__device__ unsigned long long g_lf_prof[8]; /* SVT_HIP_LF_PROFILE: cycles per stage, thread 0 of every workgroup */

constexpr int YS = 76, YROWS = 72;   /* luma tile stride / rows (8 halo + 64) */
constexpr int CS = 44, CROWS = 40;   /* chroma tile stride / rows (8 halo + 32) */

__device__ __forceinline__ int iabs(int v) { return v < 0 ? -v : v; }
__device__ __forceinline__ int sclamp(int t) { return t < -128 ? -128 : t > 127 ? 127 : t; }

/* One sample position across an edge (VPX/loopfilter.c filter4 / filter8 / filter16 with their masks), on registers:
 * p[0] = p0 (nearest the edge) ... p[7] = p7, q[0] = q0 ... q[7] = q7.  kind 4 and 8 only look at p[0..3], q[0..3]. */
__device__ __forceinline__ void filter_regs(int (&p)[8], int (&q)[8], int kind, uint32_t th) {
    const int mblim = (int)(th & 0xff), lim = (int)((th >> 8) & 0xff), hev_thr = (int)((th >> 16) & 0xff);
    const int p3 = p[3], p2 = p[2], p1 = p[1], p0 = p[0], q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
    const bool mask = !(iabs(p3 - p2) > lim || iabs(p2 - p1) > lim || iabs(p1 - p0) > lim || iabs(q1 - q0) > lim ||
                        iabs(q2 - q1) > lim || iabs(q3 - q2) > lim || iabs(p0 - q0) * 2 + iabs(p1 - q1) / 2 > mblim);
    bool flat = false;
    if (kind >= 8)
        flat = !(iabs(p1 - p0) > 1 || iabs(q1 - q0) > 1 || iabs(p2 - p0) > 1 || iabs(q2 - q0) > 1 || iabs(p3 - p0) > 1 || iabs(q3 - q0) > 1);
    if (kind == 16 && flat && mask) {
        const int p7 = p[7], p6 = p[6], p5 = p[5], p4 = p[4], q4 = q[4], q5 = q[5], q6 = q[6], q7 = q[7];
        const bool flat2 = !(iabs(p4 - p0) > 1 || iabs(q4 - q0) > 1 || iabs(p5 - p0) > 1 || iabs(q5 - q0) > 1 || iabs(p6 - p0) > 1 ||
                             iabs(q6 - q0) > 1 || iabs(p7 - p0) > 1 || iabs(q7 - q0) > 1);
        if (flat2) {
#define R4(x) (((x) + 8) >> 4)
            p[6] = R4(p7 * 7 + p6 * 2 + p5 + p4 + p3 + p2 + p1 + p0 + q0);
            p[5] = R4(p7 * 6 + p6 + p5 * 2 + p4 + p3 + p2 + p1 + p0 + q0 + q1);
            p[4] = R4(p7 * 5 + p6 + p5 + p4 * 2 + p3 + p2 + p1 + p0 + q0 + q1 + q2);
            p[3] = R4(p7 * 4 + p6 + p5 + p4 + p3 * 2 + p2 + p1 + p0 + q0 + q1 + q2 + q3);
            p[2] = R4(p7 * 3 + p6 + p5 + p4 + p3 + p2 * 2 + p1 + p0 + q0 + q1 + q2 + q3 + q4);
            p[1] = R4(p7 * 2 + p6 + p5 + p4 + p3 + p2 + p1 * 2 + p0 + q0 + q1 + q2 + q3 + q4 + q5);
            p[0] = R4(p7 + p6 + p5 + p4 + p3 + p2 + p1 + p0 * 2 + q0 + q1 + q2 + q3 + q4 + q5 + q6);
            q[0] = R4(p6 + p5 + p4 + p3 + p2 + p1 + p0 + q0 * 2 + q1 + q2 + q3 + q4 + q5 + q6 + q7);
            q[1] = R4(p5 + p4 + p3 + p2 + p1 + p0 + q0 + q1 * 2 + q2 + q3 + q4 + q5 + q6 + q7 * 2);
            q[2] = R4(p4 + p3 + p2 + p1 + p0 + q0 + q1 + q2 * 2 + q3 + q4 + q5 + q6 + q7 * 3);
            q[3] = R4(p3 + p2 + p1 + p0 + q0 + q1 + q2 + q3 * 2 + q4 + q5 + q6 + q7 * 4);
            q[4] = R4(p2 + p1 + p0 + q0 + q1 + q2 + q3 + q4 * 2 + q5 + q6 + q7 * 5);
            q[5] = R4(p1 + p0 + q0 + q1 + q2 + q3 + q4 + q5 * 2 + q6 + q7 * 6);
            q[6] = R4(p0 + q0 + q1 + q2 + q3 + q4 + q5 + q6 * 2 + q7 * 7);
#undef R4
            return;
        }
    }
    if (flat && mask) {
#define R3(x) (((x) + 4) >> 3)
        p[2] = R3(p3 + p3 + p3 + 2 * p2 + p1 + p0 + q0);
        p[1] = R3(p3 + p3 + p2 + 2 * p1 + p0 + q0 + q1);
        p[0] = R3(p3 + p2 + p1 + 2 * p0 + q0 + q1 + q2);
        q[0] = R3(p2 + p1 + p0 + 2 * q0 + q1 + q2 + q3);
        q[1] = R3(p1 + p0 + q0 + 2 * q1 + q2 + q3 + q3);
        q[2] = R3(p0 + q0 + q1 + 2 * q2 + q3 + q3 + q3);
#undef R3
        return;
    }
    /* filter4: signed 8-bit arithmetic */
    const int m = mask ? -1 : 0;
    const int hev = (iabs(p1 - p0) > hev_thr || iabs(q1 - q0) > hev_thr) ? -1 : 0;
    const int ps1 = (int8_t)(p1 ^ 0x80), ps0 = (int8_t)(p0 ^ 0x80), qs0 = (int8_t)(q0 ^ 0x80), qs1 = (int8_t)(q1 ^ 0x80);
    int f = sclamp(ps1 - qs1) & hev;
    f = sclamp(f + 3 * (qs0 - ps0)) & m;
    const int f1 = sclamp(f + 4) >> 3, f2 = sclamp(f + 3) >> 3;
    q[0] = (uint8_t)(sclamp(qs0 - f1) ^ 0x80);
    p[0] = (uint8_t)(sclamp(ps0 + f2) ^ 0x80);
    f = ((f1 + 1) >> 1) & ~hev;
    q[1] = (uint8_t)(sclamp(qs1 - f) ^ 0x80);
    p[1] = (uint8_t)(sclamp(ps1 + f) ^ 0x80);
}

/* Edge descriptor of one 8x8 block along the filtering direction: which filters run on its leading edge and on
 * its inner 4-sample edge, with the levels they use (built once per SB from the masks). */
enum { LF_K16 = 1, LF_K8 = 2, LF_K4 = 4, LF_KI = 8 };
__device__ __forceinline__ uint32_t lf_entry(int flags, int lvl16, int lvl84, int lvli) {
    return (uint32_t)flags | ((uint32_t)lvl16 << 8) | ((uint32_t)lvl84 << 14) | ((uint32_t)lvli << 20);
}

/* all filters of one block: leading edge between p and q, then the inner edge inside q (q[3..0] | q[4..7]) */
__device__ __forceinline__ void lf_block_edges(int (&p)[8], int (&q)[8], uint32_t e, const uint32_t *thr) {
    if (e & LF_K16) filter_regs(p, q, 16, thr[(e >> 8) & 63]);
    if (e & LF_K8) filter_regs(p, q, 8, thr[(e >> 14) & 63]);
    if (e & LF_K4) filter_regs(p, q, 4, thr[(e >> 14) & 63]);
}
__device__ __forceinline__ void lf_inner_edge(int (&q)[8], uint32_t e, const uint32_t *thr) {
    if (e & LF_KI) {
        int ip[8] = {q[3], q[2], q[1], q[0], 0, 0, 0, 0}, iq[8] = {q[4], q[5], q[6], q[7], 0, 0, 0, 0};
        filter_regs(ip, iq, 4, thr[(e >> 20) & 63]);
        q[3] = ip[0]; q[2] = ip[1]; q[4] = iq[0]; q[5] = iq[1];
    }
}

/* eb_vp9_adjust_mask, vp9_loopfilter.c:786-900 */
__device__ __forceinline__ void adjust_mask(svt_lf_mask &m, int mi_row, int mi_col, int mi_rows, int mi_cols) {
    m.left_y[2] |= m.left_y[3]; m.above_y[2] |= m.above_y[3];
    m.left_uv[2] |= m.left_uv[3]; m.above_uv[2] |= m.above_uv[3];
    m.left_y[1] |= m.left_y[0] & 0x1111111111111111ULL; m.left_y[0] &= ~0x1111111111111111ULL;
    m.above_y[1] |= m.above_y[0] & 0x000000ff000000ffULL; m.above_y[0] &= ~0x000000ff000000ffULL;
    m.left_uv[1] |= m.left_uv[0] & 0x1111; m.left_uv[0] &= (uint16_t)~0x1111;
    m.above_uv[1] |= m.above_uv[0] & 0x000f; m.above_uv[0] &= (uint16_t)~0x000f;
    if (mi_row + 8 > mi_rows) {
        const uint64_t rows = (uint64_t)(mi_rows - mi_row);
        const uint64_t my = (((uint64_t)1 << (rows << 3)) - 1);
        const uint16_t muv = (uint16_t)(((uint16_t)1 << (((rows + 1) >> 1) << 2)) - 1);
        for (int i = 0; i < 3; i++) { m.left_y[i] &= my; m.above_y[i] &= my; m.left_uv[i] &= muv; m.above_uv[i] &= muv; }
        m.int_4x4_y &= my; m.int_4x4_uv &= muv;
        if (rows == 1) { m.above_uv[1] |= m.above_uv[2]; m.above_uv[2] = 0; }
        if (rows == 5) { m.above_uv[1] |= m.above_uv[2] & 0xff00; m.above_uv[2] &= (uint16_t)~(m.above_uv[2] & 0xff00); }
    }
    if (mi_col + 8 > mi_cols) {
        const uint64_t cols = (uint64_t)(mi_cols - mi_col);
        const uint64_t my = (uint64_t)((1 << cols) - 1) * 0x0101010101010101ULL;
        const uint16_t muv = (uint16_t)(((1 << ((cols + 1) >> 1)) - 1) * 0x1111);
        const uint16_t muvi = (uint16_t)(((1 << (cols >> 1)) - 1) * 0x1111);
        for (int i = 0; i < 3; i++) { m.left_y[i] &= my; m.above_y[i] &= my; m.left_uv[i] &= muv; m.above_uv[i] &= muv; }
        m.int_4x4_y &= my; m.int_4x4_uv &= muvi;
        if (cols == 1) { m.left_uv[1] |= m.left_uv[2]; m.left_uv[2] = 0; }
        if (cols == 5) { m.left_uv[1] |= (m.left_uv[2] & 0xcccc); m.left_uv[2] &= (uint16_t)~(m.left_uv[2] & 0xcccc); }
    }
    if (mi_col == 0)
        for (int i = 0; i < 3; i++) { m.left_y[i] &= 0xfefefefefefefefeULL; m.left_uv[i] &= 0xeeee; }
}

/* vertical edges (filter_selectively_vert_row2, vp9_loopfilter.c:305-397) of the 8-row band `half` of a row pair,
 * column block c: m16/m8/m4/mi4 = the pair's masks (bit c = upper band, bit c+fwd = lower band), lfl0/lfl1 = levels
 * of the two bands.  [quirk] a 16-wide edge present in both bands uses the upper band's thresholds for both
 * (vp9_loopfilter.c:325-327). */
__device__ __forceinline__ uint32_t vert_entry(int c, int fwd, unsigned m16, unsigned m8, unsigned m4, unsigned mi4, int half, int lfl0, int lfl1) {
    const unsigned b0 = 1u << c, b1 = 1u << (c + fwd), mine = half ? b1 : b0;
    const int      own = half ? lfl1 : lfl0;
    int            flags = 0, l16 = own;
    if (m16 & mine) { flags |= LF_K16; if ((m16 & b0) && (m16 & b1)) l16 = lfl0; }
    if (m8 & mine) flags |= LF_K8;
    if (m4 & mine) flags |= LF_K4;
    if (mi4 & mine) flags |= LF_KI;
    return lf_entry(flags, l16, own, own);
}

/* horizontal edges of one 8-row band, column block cb: replays the greedy left-to-right pairing of
 * filter_selectively_horiz (vp9_loopfilter.c:481-568) to find this block's filter width and thresholds.
 * [quirk] the second block of a 16-wide pair uses the first block's thresholds (vp9_loopfilter.c:492-494). */
__device__ __forceinline__ uint32_t horiz_entry(int cb, unsigned m16, unsigned m8, unsigned m4, unsigned mi4, const uint8_t *lfl, int lstep) {
    int  kind = 0, lvl = 0, ilvl = 0;
    bool inner = false;
    int  p = 0;
    while (p <= cb) {
        const unsigned any = (m16 | m8 | m4 | mi4) >> p;
        if (!any) break;
        int count = 1;
        if (any & 1) {
            const int l0 = lfl[p * lstep];
            if ((m16 >> p) & 1) {
                const bool pair = ((m16 >> p) & 3) == 3;
                if (pair) count = 2;
                if (p == cb || (pair && p + 1 == cb)) { kind = LF_K16; lvl = l0; }
            } else if (((m8 >> p) & 1) || ((m4 >> p) & 1)) {
                const int      k  = ((m8 >> p) & 1) ? LF_K8 : LF_K4;
                const unsigned mk = ((m8 >> p) & 1) ? m8 : m4;
                if (((mk >> p) & 3) == 3) {
                    count = 2;
                    if (p == cb) { kind = k; lvl = l0; inner = (mi4 >> p) & 1; ilvl = l0; }
                    else if (p + 1 == cb) { kind = k; lvl = lfl[(p + 1) * lstep]; inner = (mi4 >> (p + 1)) & 1; ilvl = lvl; }
                } else if (p == cb) { kind = k; lvl = l0; inner = (mi4 >> p) & 1; ilvl = l0; }
            } else if (p == cb) { inner = true; ilvl = l0; }
        }
        p += count;
    }
    return lf_entry(kind | (inner ? LF_KI : 0), lvl, lvl, ilvl);
}

/* 8 consecutive samples <-> registers.  Along a tile row (vertical edges) they are two aligned dwords. */
__device__ __forceinline__ void row_load8(const uint8_t *s, int (&v)[8], bool reversed) {
    const uint32_t *w = (const uint32_t *)s;
    const uint32_t  a = w[0], b = w[1];
    int t[8] = {(int)(a & 0xff), (int)((a >> 8) & 0xff), (int)((a >> 16) & 0xff), (int)(a >> 24),
                (int)(b & 0xff), (int)((b >> 8) & 0xff), (int)((b >> 16) & 0xff), (int)(b >> 24)};
    _Pragma("unroll") for (int i = 0; i < 8; i++) v[i] = reversed ? t[7 - i] : t[i];
}
__device__ __forceinline__ void row_store8(uint8_t *s, const int (&v)[8], bool reversed) {
    int t[8];
    _Pragma("unroll") for (int i = 0; i < 8; i++) t[i] = reversed ? v[7 - i] : v[i];
    uint32_t *w = (uint32_t *)s;
    w[0] = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
    w[1] = (uint32_t)t[4] | ((uint32_t)t[5] << 8) | ((uint32_t)t[6] << 16) | ((uint32_t)t[7] << 24);
}
/* along a tile column (horizontal edges): 8 byte accesses at the tile stride */
__device__ __forceinline__ void col_load8(const uint8_t *s, int st, int (&v)[8], bool reversed) {
    _Pragma("unroll") for (int i = 0; i < 8; i++) v[reversed ? 7 - i : i] = s[i * st];
}
__device__ __forceinline__ void col_store8(uint8_t *s, int st, const int (&v)[8], bool reversed) {
    _Pragma("unroll") for (int i = 0; i < 8; i++) s[i * st] = (uint8_t)v[reversed ? 7 - i : i];
}

/* One line of samples (a tile row for the vertical edges, a tile column for the horizontal ones) through its nblk
 * blocks: p slides along in registers, every sample is read and written once.  s = first sample of block 0 (the 8
 * samples before it are the neighbour's), tab[c * tstep] = descriptor of this line's block c. */
template <bool ROW>
__device__ __forceinline__ void lf_line(uint8_t *s, int st, int nblk, const uint32_t *tab, int tstep, const uint32_t *thr) {
    int p[8], q[8];
    if (ROW) row_load8(s - 8, p, true); else col_load8(s - 8 * st, st, p, true);
    for (int c = 0; c < nblk; c++) {
        const uint32_t e = tab[c * tstep];
        if (ROW) row_load8(s + 8 * c, q, false); else col_load8(s + 8 * c * st, st, q, false);
        lf_block_edges(p, q, e, thr);
        if (ROW) row_store8(s + 8 * c - 8, p, true); else col_store8(s + (8 * c - 8) * st, st, p, true);
        lf_inner_edge(q, e, thr);
        _Pragma("unroll") for (int i = 0; i < 8; i++) p[i] = q[7 - i];
    }
    if (ROW) row_store8(s + 8 * nblk - 8, p, true); else col_store8(s + (8 * nblk - 8) * st, st, p, true);
}

/* copy a tile between global memory and LDS, clipped to the part of the plane that exists.
 * tile sample (tx, ty) <-> plane sample (x0 + tx, y0 + ty); only tx in [0,nx), ty in [0,ny).  x0, nx and the LDS
 * offsets are multiples of 8; when the plane rows are 8-byte aligned the copy moves 8 bytes per lane and keeps up to
 * UNITS loads in flight before the first store (a tile is a handful of units per lane). */
template <int UNITS>
__device__ __forceinline__ void tile_io(bool load, uint8_t *g, int gstride, uint8_t *l, int lstride, int x0, int y0, int nx, int ny,
                                        int tid, int nthreads) {
    if ((((uintptr_t)g | (uintptr_t)gstride) & 7) == 0) {
        const int nu = nx >> 3, total = nu * ny; /* 8-byte units per row */
        for (int t0 = tid; t0 < total; t0 += UNITS * nthreads) {
            uint2 v[UNITS];
            int   lo[UNITS];
            _Pragma("unroll") for (int u = 0; u < UNITS; u++) {
                const int t = t0 + u * nthreads;
                lo[u] = -1;
                if (t < total) {
                    const int ty = t / nu, tu = t - ty * nu;
                    uint2    *gp = (uint2 *)(g + (ptrdiff_t)(y0 + ty) * gstride + x0 + 8 * tu);
                    lo[u] = ty * lstride + 8 * tu;
                    if (load) v[u] = *gp;
                    else { const uint32_t *lp = (const uint32_t *)(l + lo[u]); *gp = make_uint2(lp[0], lp[1]); }
                }
            }
            if (load) {
                _Pragma("unroll") for (int u = 0; u < UNITS; u++)
                    if (lo[u] >= 0) { uint32_t *lp = (uint32_t *)(l + lo[u]); lp[0] = v[u].x; lp[1] = v[u].y; }
            }
        }
        return;
    }
    for (int t = tid; t < nx * ny; t += nthreads) {
        const int ty = t / nx, tx = t - ty * nx;
        uint8_t  *gp = g + (ptrdiff_t)(y0 + ty) * gstride + x0 + tx;
        if (load) l[ty * lstride + tx] = *gp;
        else *gp = l[ty * lstride + tx];
    }
}

__global__ __launch_bounds__(128) void svt_lf_kernel(const lf_pic_dev *__restrict__ pics, int n_pics, svt_lf_thresh thr,
                                                     uint32_t *__restrict__ ticket, int rows_per_pic, int prof) {
    __shared__ __align__(16) uint8_t ytile[YROWS * YS];
    __shared__ __align__(16) uint8_t ctile[2][CROWS * CS];
    __shared__ uint32_t              s_thr[64];      /* mblim | lim << 8 | hev_thr << 16 per filter level */
    __shared__ uint32_t              s_vy[64], s_hy[64]; /* luma edge descriptors [8-row band][column block] */
    __shared__ uint32_t              s_vc[16], s_hc[16]; /* chroma */
    __shared__ int                   s_job;
    const int tid = threadIdx.x;
    if (tid == 0) s_job = (int)atomicAdd(ticket, 1u);
    if (tid < 64) s_thr[tid] = (uint32_t)thr.mblim[tid] | ((uint32_t)thr.lim[tid] << 8) | ((uint32_t)thr.hev_thr[tid] << 16);
    __syncthreads();
    const int job = s_job;
    if (job >= n_pics * rows_per_pic) return;
    const lf_pic_dev P = pics[job / rows_per_pic];
    const int sb_row = job % rows_per_pic;
    const int sb_cols = (P.mi_cols + 7) >> 3, sb_rows = (P.mi_rows + 7) >> 3;
    if (sb_row >= sb_rows) return;
    const int W = P.planes.width, H = P.planes.height, CW = W >> 1, CH = H >> 1;
    const int mi_row = sb_row * 8;
    const int wave = tid >> 6, lane = tid & 63;
    const int nrows = P.mi_rows - mi_row < 8 ? P.mi_rows - mi_row : 8; /* 8-row bands of this SB row */

    unsigned long long tm_ = prof ? __builtin_amdgcn_s_memtime() : 0;
#define LF_MARK(i) do { if (prof && tid == 0) { unsigned long long n_ = __builtin_amdgcn_s_memtime(); atomicAdd(&g_lf_prof[i], n_ - tm_); tm_ = n_; } } while (0)
    for (int sc = 0; sc < sb_cols; sc++) {
        /* ---- wait for (sb_row-1, sc+1) ---- */
        if (sb_row > 0) {
            const uint32_t need = (uint32_t)(sc + 2 < sb_cols ? sc + 2 : sb_cols);
            if (tid == 0) {
                while (__hip_atomic_load(&P.progress[sb_row - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) __builtin_amdgcn_s_sleep(1);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            __syncthreads();
        }
        LF_MARK(0);
        const int mi_col = sc * 8;
        const int x0 = sc * 64, y0 = sb_row * 64, cx0 = sc * 32, cy0 = sb_row * 32;
        /* tile extents that exist in the planes (halo of 8 up/left where there is a neighbour) */
        const int hx = sc > 0 ? 8 : 0, hy = sb_row > 0 ? 8 : 0;
        const int vw = (W - x0) < 64 ? W - x0 : 64, vh = (H - y0) < 64 ? H - y0 : 64;
        const int cvw = (CW - cx0) < 32 ? CW - cx0 : 32, cvh = (CH - cy0) < 32 ? CH - cy0 : 32;
        uint8_t *yl = ytile + (8 - hy) * YS + (8 - hx); /* LDS address of tile sample (x0-hx, y0-hy) */
        tile_io<6>(true, P.planes.y, P.planes.y_stride, yl, YS, x0 - hx, y0 - hy, vw + hx, vh + hy, tid, 128);
        if (!P.y_only) {
            tile_io<2>(true, P.planes.u, P.planes.uv_stride, ctile[0] + (8 - hy) * CS + (8 - hx), CS, cx0 - hx, cy0 - hy, cvw + hx, cvh + hy, tid, 128);
            tile_io<2>(true, P.planes.v, P.planes.uv_stride, ctile[1] + (8 - hy) * CS + (8 - hx), CS, cx0 - hx, cy0 - hy, cvw + hx, cvh + hy, tid, 128);
        }
        LF_MARK(1);
        /* ---- edge descriptors of this SB: one (band, block) per lane ---- */
        {
            svt_lf_mask m = P.lfm[sb_row * P.lfm_stride + sc];
            adjust_mask(m, mi_row, mi_col, P.mi_rows, P.mi_cols);
            if (tid < 64) {
                const int rr = tid >> 3, c = tid & 7, pair = rr >> 1, half = rr & 1;
                s_vy[tid] = vert_entry(c, 8, (unsigned)(m.left_y[2] >> (16 * pair)) & 0xffff, (unsigned)(m.left_y[1] >> (16 * pair)) & 0xffff,
                                       (unsigned)(m.left_y[0] >> (16 * pair)) & 0xffff, (unsigned)(m.int_4x4_y >> (16 * pair)) & 0xffff, half,
                                       m.lfl_y[(2 * pair) * 8 + c], m.lfl_y[(2 * pair + 1) * 8 + c]);
                const int r = rr;
                unsigned  a16 = 0, a8 = 0, a4 = 0;
                if (mi_row + r != 0) { a16 = (unsigned)(m.above_y[2] >> (8 * r)) & 0xff; a8 = (unsigned)(m.above_y[1] >> (8 * r)) & 0xff; a4 = (unsigned)(m.above_y[0] >> (8 * r)) & 0xff; }
                s_hy[tid] = horiz_entry(c, a16, a8, a4, (unsigned)(m.int_4x4_y >> (8 * r)) & 0xff, &m.lfl_y[r * 8], 1);
            } else if (tid < 80) {
                const int t = tid - 64, rr = t >> 2, c = t & 3, pair = rr >> 1, half = rr & 1;
                s_vc[t] = vert_entry(c, 4, (unsigned)(m.left_uv[2] >> (8 * pair)) & 0xff, (unsigned)(m.left_uv[1] >> (8 * pair)) & 0xff,
                                     (unsigned)(m.left_uv[0] >> (8 * pair)) & 0xff, (unsigned)(m.int_4x4_uv >> (8 * pair)) & 0xff, half,
                                     m.lfl_y[(4 * pair) * 8 + 2 * c], m.lfl_y[(4 * pair + 2) * 8 + 2 * c]);
            } else if (tid < 96) {
                const int t = tid - 80, ru = t >> 2, c = t & 3, r = 2 * ru;
                unsigned  a16 = 0, a8 = 0, a4 = 0;
                if (mi_row + r != 0) { a16 = (unsigned)(m.above_uv[2] >> (4 * ru)) & 0xf; a8 = (unsigned)(m.above_uv[1] >> (4 * ru)) & 0xf; a4 = (unsigned)(m.above_uv[0] >> (4 * ru)) & 0xf; }
                const unsigned mi4 = (mi_row + r == P.mi_rows - 1) ? 0u : ((unsigned)(m.int_4x4_uv >> (4 * ru)) & 0xf);
                s_hc[t] = horiz_entry(c, a16, a8, a4, mi4, &m.lfl_y[r * 8], 2);
            }
        }
        __syncthreads();
        LF_MARK(2);
        /* ---- vertical edges: lane = sample row ---- */
        if (wave == 0) {
            if (lane < vh) lf_line<true>(ytile + (8 + lane) * YS + 8, 1, 8, &s_vy[(lane >> 3) * 8], 1, s_thr);
        } else if (!P.y_only) {
            const int pl = lane >> 5, r = lane & 31;
            if (r < cvh) lf_line<true>(ctile[pl] + (8 + r) * CS + 8, 1, 4, &s_vc[(r >> 3) * 4], 1, s_thr);
        }
        __syncthreads();
        LF_MARK(3);
        /* ---- horizontal edges: lane = sample column; a band's descriptor sits at [band*8 + column block] ---- */
        if (wave == 0) {
            if (lane < vw) lf_line<false>(ytile + 8 * YS + 8 + lane, YS, nrows, &s_hy[lane >> 3], 8, s_thr);
        } else if (!P.y_only) {
            const int pl = lane >> 5, x = lane & 31;
            if (x < cvw) lf_line<false>(ctile[pl] + 8 * CS + 8 + x, CS, (nrows + 1) >> 1, &s_hc[x >> 3], 4, s_thr);
        }
        __syncthreads();
        LF_MARK(4);
        /* ---- write back ---- */
        tile_io<6>(false, P.planes.y, P.planes.y_stride, yl, YS, x0 - hx, y0 - hy, vw + hx, vh + hy, tid, 128);
        if (!P.y_only) {
            tile_io<2>(false, P.planes.u, P.planes.uv_stride, ctile[0] + (8 - hy) * CS + (8 - hx), CS, cx0 - hx, cy0 - hy, cvw + hx, cvh + hy, tid, 128);
            tile_io<2>(false, P.planes.v, P.planes.uv_stride, ctile[1] + (8 - hy) * CS + (8 - hx), CS, cx0 - hx, cy0 - hy, cvw + hx, cvh + hy, tid, 128);
        }
        /* ---- publish: all stores of this workgroup -> agent-scope release -> progress counter ---- */
        __syncthreads();
        LF_MARK(5);
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(&P.progress[sb_row], (uint32_t)(sc + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        LF_MARK(6);
    }
#undef LF_MARK
}
} // namespace

static int32_t lf_launch(svt_hip_ctx *ctx, int n_pics, const svt_yuv_planes *d_recon, const svt_lf_mask *const *d_lfm, const int32_t *lfm_stride,
                         const svt_lf_thresh *thr, const int32_t *mi_rows, const int32_t *mi_cols, int32_t y_only) {
    int max_rows = 0;
    for (int i = 0; i < n_pics; i++) {
        if (mi_rows[i] < 1 || mi_cols[i] < 1 || d_recon[i].width != mi_cols[i] * 8 || d_recon[i].height != mi_rows[i] * 8)
            return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "lf: plane size must be mi_cols*8 x mi_rows*8");
        const int r = (mi_rows[i] + 7) / 8;
        if (r > max_rows) max_rows = r;
    }
    HIP_TRY(hipSetDevice(ctx->device));
    /* descriptors + progress counters + ticket in one device scratch block */
    const size_t desc_bytes = (sizeof(lf_pic_dev) * (size_t)n_pics + 15) & ~(size_t)15;
    const size_t cnt_words  = (size_t)n_pics * max_rows + 4;
    lf_pic_dev  *h = nullptr;
    uint8_t     *d = nullptr;
    if (svt_ctx_stage(ctx, desc_bytes + cnt_words * 4, (void **)&h, (void **)&d)) return svt_set_error(SVT_HIP_ERR_NO_RESOURCES, "lf: scratch");
    uint32_t *cnt = (uint32_t *)(d + desc_bytes);
    for (int i = 0; i < n_pics; i++) {
        h[i].planes = d_recon[i]; h[i].lfm = d_lfm[i]; h[i].lfm_stride = lfm_stride[i]; h[i].mi_rows = mi_rows[i]; h[i].mi_cols = mi_cols[i];
        h[i].y_only = y_only; h[i].progress = cnt + 4 + (size_t)i * max_rows;
    }
    HIP_TRY(hipMemcpyAsync(d, h, sizeof(lf_pic_dev) * (size_t)n_pics, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemsetAsync(cnt, 0, cnt_words * 4, ctx->stream));
    HIP_TRY(hipEventRecord(ctx->ev_start, ctx->stream));
    static const bool want_prof = getenv("SVT_HIP_LF_PROFILE") != nullptr;
    if (want_prof) { unsigned long long z[8] = {0}; HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_lf_prof), z, sizeof z)); }
    hipLaunchKernelGGL(svt_lf_kernel, dim3(n_pics * max_rows), dim3(128), 0, ctx->stream, (const lf_pic_dev *)d, n_pics, *thr, cnt, max_rows, want_prof ? 1 : 0);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(ctx->ev_stop, ctx->stream));
    if (want_prof) {
        unsigned long long hp[8];
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        HIP_TRY(hipMemcpyFromSymbol(hp, HIP_SYMBOL(g_lf_prof), sizeof hp));
        static const char *nm[7] = {"wait", "tile_load", "descriptors", "vertical", "horizontal", "tile_store", "publish"};
        unsigned long long tot = 0, steps = 0;
        for (int i = 0; i < 7; i++) tot += hp[i];
        for (int i = 0; i < n_pics; i++) steps += (unsigned long long)((mi_rows[i] + 7) / 8) * ((mi_cols[i] + 7) / 8);
        fprintf(stderr, "[lf-profile] pics=%d SB steps=%llu avg cycles/SB=%llu :", n_pics, steps, tot / (steps ? steps : 1));
        for (int i = 0; i < 7; i++) fprintf(stderr, " %s=%.1f%%", nm[i], 100.0 * (double)hp[i] / (double)tot);
        fprintf(stderr, "\n");
    }
    svt_ctx_stage_commit(ctx);
    ctx->timed = 1;
    return SVT_HIP_OK;
}

extern "C" int32_t svt_hip_lf_frame_device(svt_hip_ctx *ctx, const svt_yuv_planes *d_recon, const svt_lf_mask *d_lfm, int32_t lfm_stride,
                                           const svt_lf_thresh *thr, int32_t mi_rows, int32_t mi_cols, int32_t y_only) {
    if (!ctx || !d_recon || !d_lfm || !thr) return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "lf: null argument");
    return lf_launch(ctx, 1, d_recon, &d_lfm, &lfm_stride, thr, &mi_rows, &mi_cols, y_only);
}

extern "C" int32_t svt_hip_lf_batch_device(svt_hip_ctx *ctx, int32_t n_pics, const svt_yuv_planes *d_recon, const svt_lf_mask *const *d_lfm,
                                           const int32_t *lfm_stride, const svt_lf_thresh *thr, const int32_t *mi_rows,
                                           const int32_t *mi_cols, int32_t y_only) {
    if (!ctx || n_pics < 1 || !d_recon || !d_lfm || !lfm_stride || !thr || !mi_rows || !mi_cols)
        return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "lf: null argument");
    return lf_launch(ctx, n_pics, d_recon, d_lfm, lfm_stride, thr, mi_rows, mi_cols, y_only);
}

extern "C" int32_t svt_hip_lf_frame(svt_hip_ctx *ctx, const svt_yuv_planes *recon, const svt_lf_mask *lfm, int32_t lfm_stride,
                                    const svt_lf_thresh *thr, int32_t mi_rows, int32_t mi_cols, int32_t y_only) {
    if (!ctx || !recon || !lfm || !thr || !recon->y || (!y_only && (!recon->u || !recon->v)))
        return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "lf: null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    const int    H = recon->height, CH = H / 2;
    const size_t yb = (size_t)recon->y_stride * H, cb = (size_t)recon->uv_stride * CH;
    const int    sb_rows = (mi_rows + 7) / 8;
    const size_t mb = sizeof(svt_lf_mask) * (size_t)sb_rows * lfm_stride;
    uint8_t *dy = (uint8_t *)svt_ctx_slot(ctx, 20, yb + 64), *du = (uint8_t *)svt_ctx_slot(ctx, 21, cb + 64), *dv = (uint8_t *)svt_ctx_slot(ctx, 22, cb + 64);
    svt_lf_mask *dm = (svt_lf_mask *)svt_ctx_slot(ctx, 23, mb);
    if (!dy || !du || !dv || !dm) return svt_set_error(SVT_HIP_ERR_NO_RESOURCES, "lf: device buffers");
    HIP_TRY(hipMemcpyAsync(dy, recon->y, yb, hipMemcpyHostToDevice, ctx->stream));
    if (!y_only) {
        HIP_TRY(hipMemcpyAsync(du, recon->u, cb, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(hipMemcpyAsync(dv, recon->v, cb, hipMemcpyHostToDevice, ctx->stream));
    }
    HIP_TRY(hipMemcpyAsync(dm, lfm, mb, hipMemcpyHostToDevice, ctx->stream));
    svt_yuv_planes d = *recon;
    d.y = dy; d.u = du; d.v = dv;
    int32_t rc = svt_hip_lf_frame_device(ctx, &d, dm, lfm_stride, thr, mi_rows, mi_cols, y_only);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(recon->y, dy, yb, hipMemcpyDeviceToHost, ctx->stream));
    if (!y_only) {
        HIP_TRY(hipMemcpyAsync(recon->u, du, cb, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipMemcpyAsync(recon->v, dv, cb, hipMemcpyDeviceToHost, ctx->stream));
    }
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return SVT_HIP_OK;
}
