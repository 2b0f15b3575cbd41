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
#include "svt_ctx.h"

namespace {

struct lf_pic_dev {
    svt_yuv_planes     planes;
    const svt_lf_mask *lfm;
    int32_t            lfm_stride, mi_rows, mi_cols, y_only;
    uint32_t          *progress; /* [sb_rows] SBs completed per SB row */
};

constexpr int YS = 76, YROWS = 72;   /* luma tile stride / rows (8 halo + 64) */
constexpr int CS = 44, CROWS = 40;   /* chroma tile stride / rows (8 halo + 32) */

__device__ __forceinline__ int iabs(int v) { return v < 0 ? -v : v; }
__device__ __forceinline__ int sclamp(int t) { return t < -128 ? -128 : t > 127 ? 127 : t; }

/* one sample position across an edge (VPX/loopfilter.c filter4 / filter8 / filter16 with their masks) */
__device__ __forceinline__ void filter_px(uint8_t *s, int st, int kind, int mblim, int lim, int hev_thr) {
    const int p3 = s[-4 * st], p2 = s[-3 * st], p1 = s[-2 * st], p0 = s[-st];
    const int q0 = s[0], q1 = s[st], q2 = s[2 * st], q3 = s[3 * st];
    const bool mask = !(iabs(p3 - p2) > lim || iabs(p2 - p1) > lim || iabs(p1 - p0) > lim || iabs(q1 - q0) > lim ||
                        iabs(q2 - q1) > lim || iabs(q3 - q2) > lim || iabs(p0 - q0) * 2 + iabs(p1 - q1) / 2 > mblim);
    bool flat = false;
    if (kind >= 8)
        flat = !(iabs(p1 - p0) > 1 || iabs(q1 - q0) > 1 || iabs(p2 - p0) > 1 || iabs(q2 - q0) > 1 || iabs(p3 - p0) > 1 || iabs(q3 - q0) > 1);
    if (kind == 16 && flat && mask) {
        const int p7 = s[-8 * st], p6 = s[-7 * st], p5 = s[-6 * st], p4 = s[-5 * st];
        const int q4 = s[4 * st], q5 = s[5 * st], q6 = s[6 * st], q7 = s[7 * st];
        const bool flat2 = !(iabs(p4 - p0) > 1 || iabs(q4 - q0) > 1 || iabs(p5 - p0) > 1 || iabs(q5 - q0) > 1 || iabs(p6 - p0) > 1 ||
                             iabs(q6 - q0) > 1 || iabs(p7 - p0) > 1 || iabs(q7 - q0) > 1);
        if (flat2) {
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
    /* filter4: signed 8-bit arithmetic */
    const int m = mask ? -1 : 0;
    const int hev = (iabs(p1 - p0) > hev_thr || iabs(q1 - q0) > hev_thr) ? -1 : 0;
    const int ps1 = (int8_t)(p1 ^ 0x80), ps0 = (int8_t)(p0 ^ 0x80), qs0 = (int8_t)(q0 ^ 0x80), qs1 = (int8_t)(q1 ^ 0x80);
    int f = sclamp(ps1 - qs1) & hev;
    f = sclamp(f + 3 * (qs0 - ps0)) & m;
    const int f1 = sclamp(f + 4) >> 3, f2 = sclamp(f + 3) >> 3;
    s[0]   = (uint8_t)(sclamp(qs0 - f1) ^ 0x80);
    s[-st] = (uint8_t)(sclamp(ps0 + f2) ^ 0x80);
    f = ((f1 + 1) >> 1) & ~hev;
    s[st]      = (uint8_t)(sclamp(qs1 - f) ^ 0x80);
    s[-2 * st] = (uint8_t)(sclamp(ps1 + f) ^ 0x80);
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

/* vertical edges of one sample row (filter_selectively_vert_row2 seen from one row):
 * m16/m8/m4/mi4 = the row pair's masks (bit c = upper 8-row half, bit c+fwd = lower half), `half` = which half this
 * row is in, lfl0/lfl1 = level arrays of the two halves.  [quirk] a 16-wide edge present in both halves uses the
 * upper half's thresholds for both (vp9_loopfilter.c:325-327). */
__device__ __forceinline__ void vert_row(uint8_t *row, int nblk, int fwd, unsigned m16, unsigned m8, unsigned m4, unsigned mi4, int half,
                                         const uint8_t *lfl0, const uint8_t *lfl1, const svt_lf_thresh &t) {
    for (int c = 0; c < nblk; c++) {
        const unsigned b0 = 1u << c, b1 = 1u << (c + fwd), mine = half ? b1 : b0;
        const int      own = half ? lfl1[c] : lfl0[c];
        uint8_t       *s = row + 8 * c;
        if (m16 & mine) {
            const int l = ((m16 & b0) && (m16 & b1)) ? lfl0[c] : own;
            filter_px(s, 1, 16, t.mblim[l], t.lim[l], t.hev_thr[l]);
        }
        if (m8 & mine) filter_px(s, 1, 8, t.mblim[own], t.lim[own], t.hev_thr[own]);
        if (m4 & mine) filter_px(s, 1, 4, t.mblim[own], t.lim[own], t.hev_thr[own]);
        if (mi4 & mine) filter_px(s + 4, 1, 4, t.mblim[own], t.lim[own], t.hev_thr[own]);
    }
}

/* horizontal edges of one 8-row band seen from one sample column in column block cb: replays the greedy left-to-right
 * pairing of filter_selectively_horiz to find this block's filter width and thresholds.
 * [quirk] the second block of a 16-wide pair uses the first block's thresholds (vp9_loopfilter.c:492-494). */
__device__ __forceinline__ void horiz_col(uint8_t *s, int st, int cb, unsigned m16, unsigned m8, unsigned m4, unsigned mi4,
                                          const uint8_t *lfl, const svt_lf_thresh &t) {
    int kind = 0, lvl = 0, ilvl = 0;
    bool inner = false;
    int p = 0;
    while (p <= cb) {
        const unsigned any = (m16 | m8 | m4 | mi4) >> p;
        if (!any) break;
        int count = 1;
        if (any & 1) {
            const int l0 = lfl[p];
            if ((m16 >> p) & 1) {
                const bool pair = ((m16 >> p) & 3) == 3;
                if (pair) count = 2;
                if (p == cb || (pair && p + 1 == cb)) { kind = 16; lvl = l0; }
            } else if (((m8 >> p) & 1) || ((m4 >> p) & 1)) {
                const int      k  = ((m8 >> p) & 1) ? 8 : 4;
                const unsigned mk = ((m8 >> p) & 1) ? m8 : m4;
                if (((mk >> p) & 3) == 3) {
                    count = 2;
                    if (p == cb) { kind = k; lvl = l0; inner = (mi4 >> p) & 1; ilvl = l0; }
                    else if (p + 1 == cb) { kind = k; lvl = lfl[p + 1]; inner = (mi4 >> (p + 1)) & 1; ilvl = lfl[p + 1]; }
                } else if (p == cb) { kind = k; lvl = l0; inner = (mi4 >> p) & 1; ilvl = l0; }
            } else if (p == cb) { inner = true; ilvl = l0; }
        }
        p += count;
    }
    if (kind) filter_px(s, st, kind, t.mblim[lvl], t.lim[lvl], t.hev_thr[lvl]);
    if (inner) filter_px(s + 4 * st, st, 4, t.mblim[ilvl], t.lim[ilvl], t.hev_thr[ilvl]);
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
                                                     uint32_t *__restrict__ ticket, int rows_per_pic) {
    __shared__ __align__(16) uint8_t ytile[YROWS * YS];
    __shared__ __align__(16) uint8_t ctile[2][CROWS * CS];
    __shared__ int                   s_job;
    const int tid = threadIdx.x;
    if (tid == 0) s_job = (int)atomicAdd(ticket, 1u);
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

    for (int sc = 0; sc < sb_cols; sc++) {
        /* ---- wait for (sb_row-1, sc+1) ---- */
        if (sb_row > 0) {
            const uint32_t need = (uint32_t)(sc + 2 < sb_cols ? sc + 2 : sb_cols);
            if (tid == 0) {
                while (__hip_atomic_load(&P.progress[sb_row - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) __builtin_amdgcn_s_sleep(2);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            __syncthreads();
        }
        const int mi_col = sc * 8;
        svt_lf_mask m = P.lfm[sb_row * P.lfm_stride + sc];
        adjust_mask(m, mi_row, mi_col, P.mi_rows, P.mi_cols);
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
        __syncthreads();
        /* ---- vertical edges ---- */
        if (wave == 0) {
            const int r = lane;
            if (r < vh) {
                const int rr = r >> 3, pair = rr >> 1, half = rr & 1;
                vert_row(ytile + (8 + r) * YS + 8, 8, 8, (unsigned)(m.left_y[2] >> (16 * pair)) & 0xffff, (unsigned)(m.left_y[1] >> (16 * pair)) & 0xffff,
                         (unsigned)(m.left_y[0] >> (16 * pair)) & 0xffff, (unsigned)(m.int_4x4_y >> (16 * pair)) & 0xffff, half,
                         &m.lfl_y[(2 * pair) * 8], &m.lfl_y[(2 * pair + 1) * 8], thr);
            }
        } else if (!P.y_only) {
            const int pl = lane >> 5, r = lane & 31;
            if (r < cvh) {
                const int rr = r >> 3, pair = rr >> 1, half = rr & 1;
                uint8_t   l0[4], l1[4];
                for (int c = 0; c < 4; c++) { l0[c] = m.lfl_y[(4 * pair) * 8 + 2 * c]; l1[c] = m.lfl_y[(4 * pair + 2) * 8 + 2 * c]; }
                vert_row(ctile[pl] + (8 + r) * CS + 8, 4, 4, (unsigned)(m.left_uv[2] >> (8 * pair)) & 0xff, (unsigned)(m.left_uv[1] >> (8 * pair)) & 0xff,
                         (unsigned)(m.left_uv[0] >> (8 * pair)) & 0xff, (unsigned)(m.int_4x4_uv >> (8 * pair)) & 0xff, half, l0, l1, thr);
            }
        }
        __syncthreads();
        /* ---- horizontal edges ---- */
        if (wave == 0) {
            const int x = lane;
            if (x < vw) {
                for (int r = 0; r < 8 && mi_row + r < P.mi_rows; r++) {
                    unsigned a16 = 0, a8 = 0, a4 = 0;
                    if (mi_row + r != 0) { a16 = (unsigned)(m.above_y[2] >> (8 * r)) & 0xff; a8 = (unsigned)(m.above_y[1] >> (8 * r)) & 0xff; a4 = (unsigned)(m.above_y[0] >> (8 * r)) & 0xff; }
                    horiz_col(ytile + (8 + 8 * r) * YS + 8 + x, YS, x >> 3, a16, a8, a4, (unsigned)(m.int_4x4_y >> (8 * r)) & 0xff, &m.lfl_y[r * 8], thr);
                }
            }
        } else if (!P.y_only) {
            const int pl = lane >> 5, x = lane & 31;
            if (x < cvw) {
                for (int r = 0; r < 8 && mi_row + r < P.mi_rows; r += 2) {
                    const int ru = r >> 1;
                    unsigned  a16 = 0, a8 = 0, a4 = 0;
                    if (mi_row + r != 0) { a16 = (unsigned)(m.above_uv[2] >> (4 * ru)) & 0xf; a8 = (unsigned)(m.above_uv[1] >> (4 * ru)) & 0xf; a4 = (unsigned)(m.above_uv[0] >> (4 * ru)) & 0xf; }
                    const unsigned mi4 = (mi_row + r == P.mi_rows - 1) ? 0u : ((unsigned)(m.int_4x4_uv >> (4 * ru)) & 0xf);
                    uint8_t        lu[4];
                    for (int c = 0; c < 4; c++) lu[c] = m.lfl_y[r * 8 + 2 * c];
                    horiz_col(ctile[pl] + (8 + 8 * ru) * CS + 8 + x, CS, x >> 3, a16, a8, a4, mi4, lu, thr);
                }
            }
        }
        __syncthreads();
        /* ---- write back ---- */
        tile_io<6>(false, P.planes.y, P.planes.y_stride, yl, YS, x0 - hx, y0 - hy, vw + hx, vh + hy, tid, 128);
        if (!P.y_only) {
            tile_io<2>(false, P.planes.u, P.planes.uv_stride, ctile[0] + (8 - hy) * CS + (8 - hx), CS, cx0 - hx, cy0 - hy, cvw + hx, cvh + hy, tid, 128);
            tile_io<2>(false, P.planes.v, P.planes.uv_stride, ctile[1] + (8 - hy) * CS + (8 - hx), CS, cx0 - hx, cy0 - hy, cvw + hx, cvh + hy, tid, 128);
        }
        /* ---- publish: all stores of this workgroup -> agent-scope release -> progress counter ---- */
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(&P.progress[sb_row], (uint32_t)(sc + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
    }
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
    hipLaunchKernelGGL(svt_lf_kernel, dim3(n_pics * max_rows), dim3(128), 0, ctx->stream, (const lf_pic_dev *)d, n_pics, *thr, cnt, max_rows);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(ctx->ev_stop, ctx->stream));
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
