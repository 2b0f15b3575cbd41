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
    const uint32_t    *desc;     /* [sb_rows][sb_cols][LF_DESC_WORDS] edge descriptors (svt_lf_desc_kernel) */
};

__device__ unsigned long long g_lf_prof[8]; /* SVT_HIP_LF_PROFILE: cycles per stage, thread 0 of every workgroup */
__device__ unsigned long long g_lf_rowts[64][8]; /* SVT_HIP_LF_PROFILE, picture 0: per SB row, s_memtime at: row taken, first halo in LDS, SB 0 filtered,
                                                    first progress published, SB 1's vertical pass done, last SB filtered, row complete */

constexpr int LF_DESC_WORDS = 320; /* two words per block: leading edge, inner edge (lf_entry2) */
constexpr int YS = 76, YROWS = 72;   /* luma tile stride / rows (8 halo + 64) */
constexpr int CS = 44, CROWS = 40;   /* chroma tile stride / rows (8 halo + 32) */

__device__ __forceinline__ int iabs(int v) { return v < 0 ? -v : v; }
__device__ __forceinline__ int sclamp(int t) { return t < -128 ? -128 : t > 127 ? 127 : t; }
__device__ __forceinline__ int uclamp(int t) { return t < 0 ? 0 : t > 255 ? 255 : t; }

/* One sample position across an edge (VPX/loopfilter.c filter_mask :31, flat_mask4/5 :44-63, hev_mask :65, filter4 :72,
 * filter8 :147, filter16 :209), on registers: p[0] = p0 (nearest the edge) ... p[7] = p7, q[0] = q0 ... q[7] = q7; kind 4
 * and 8 only look at p[0..3], q[0..3].
 * Branch-free per lane: the three candidate results are independent dependency chains (a lone wave issues one dependent
 * VALU op every ~5 cycles, so instruction-level parallelism is what shortens an edge), the 8- and 16-tap results are
 * running sums (each output = previous sum - 2 leaving taps + 2 entering taps) and are skipped wave-wide when no lane
 * of the wave needs them.  `kind` is per lane (0 = no edge here, 4, 8, 16): the lanes of a wave sit on different blocks
 * whose edges carry different filter widths, and one pass serves them all instead of one pass per width. */
__device__ __forceinline__ int lf_ad(int a, int b) { return (int)__builtin_amdgcn_sad_u8((unsigned)a, (unsigned)b, 0u); } /* |a - b|, 0..255 */
__device__ __forceinline__ int lf_max3(int a, int b, int c) { return max(max(a, b), c); }

template <int MAXK>   /* the widest filter a caller can ask for: an inner edge only ever takes the 4-wide one */
__device__ __forceinline__ void filter_regs(int (&p)[8], int (&q)[8], int kind, uint32_t th) {
    const int mblim = (int)(th & 0xff), lim = (int)((th >> 8) & 0xff), hev_thr = (int)((th >> 16) & 0xff);
    const int p3 = p[3], p2 = p[2], p1 = p[1], p0 = p[0], q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
    const int d10 = lf_ad(p1, p0), e10 = lf_ad(q1, q0);
    const bool mask = kind != 0 && max(lf_max3(lf_ad(p3, p2), lf_ad(p2, p1), d10), lf_max3(e10, lf_ad(q2, q1), lf_ad(q3, q2))) <= lim &&
                      lf_ad(p0, q0) * 2 + (lf_ad(p1, q1) >> 1) <= mblim;
    bool use_flat = false, use16 = false;
    if (MAXK >= 8 && __builtin_amdgcn_ballot_w64(kind >= 8)) {
        const bool flat = kind >= 8 && lf_max3(max(d10, e10), max(lf_ad(p2, p0), lf_ad(q2, q0)), max(lf_ad(p3, p0), lf_ad(q3, q0))) <= 1;
        use_flat = flat && mask;
        if (__builtin_amdgcn_ballot_w64(kind == 16)) {
            const int f2 = max(lf_max3(lf_ad(p[4], p0), lf_ad(p[5], p0), lf_ad(p[6], p0)), lf_max3(lf_ad(p[7], p0), lf_ad(q[4], q0), lf_ad(q[5], q0)));
            use16 = kind == 16 && use_flat && max(f2, max(lf_ad(q[6], q0), lf_ad(q[7], q0))) <= 1;
        }
    }
    /* filter4: signed 8-bit arithmetic; with mask = 0 every step yields 0 and the samples come out unchanged */
    const int m = mask ? -1 : 0;
    const int hev = max(d10, e10) > hev_thr ? -1 : 0;
    /* the reference works on samples biased by -128 (x ^ 0x80 as int8): the bias cancels in the two differences, and
     * signed_char_clamp(xs +- f) ^ 0x80 == clamp(x +- f, 0, 255) -- no conversions needed */
    int f = sclamp(p1 - q1) & hev;
    f = sclamp(f + 3 * (q0 - p0)) & m;
    const int f1 = sclamp(f + 4) >> 3, f2_ = sclamp(f + 3) >> 3;
    int       o_q0 = uclamp(q0 - f1), o_p0 = uclamp(p0 + f2_);
    f = ((f1 + 1) >> 1) & ~hev;
    int o_q1 = uclamp(q1 - f), o_p1 = uclamp(p1 + f);
    int o_p2 = p2, o_q2 = q2;
    if (MAXK >= 8 && __builtin_amdgcn_ballot_w64(use_flat && !use16)) { /* filter8 for the lanes that take it */
        int s = 3 * p3 + 2 * p2 + p1 + p0 + q0 + 4, r;
        r = s >> 3; o_p2 = (use_flat && !use16) ? r : o_p2;
        s += p1 + q1 - p3 - p2; r = s >> 3; o_p1 = (use_flat && !use16) ? r : o_p1;
        s += p0 + q2 - p3 - p1; r = s >> 3; o_p0 = (use_flat && !use16) ? r : o_p0;
        s += q0 + q3 - p3 - p0; r = s >> 3; o_q0 = (use_flat && !use16) ? r : o_q0;
        s += q1 + q3 - p2 - q0; r = s >> 3; o_q1 = (use_flat && !use16) ? r : o_q1;
        s += q2 + q3 - p1 - q1; r = s >> 3; o_q2 = (use_flat && !use16) ? r : o_q2;
    }
    if (MAXK >= 16 && __builtin_amdgcn_ballot_w64(use16)) { /* filter16 for the lanes that take it */
        const int p7 = p[7], p6 = p[6], p5 = p[5], p4 = p[4], q4 = q[4], q5 = q[5], q6 = q[6], q7 = q[7];
        int       s = 7 * p7 + 2 * p6 + p5 + p4 + p3 + p2 + p1 + p0 + q0 + 8, r;
        r = s >> 4; p[6] = use16 ? r : p6;
        s += p5 + q1 - p7 - p6; r = s >> 4; p[5] = use16 ? r : p5;
        s += p4 + q2 - p7 - p5; r = s >> 4; p[4] = use16 ? r : p4;
        s += p3 + q3 - p7 - p4; r = s >> 4; p[3] = use16 ? r : p3;
        s += p2 + q4 - p7 - p3; r = s >> 4; o_p2 = use16 ? r : o_p2;
        s += p1 + q5 - p7 - p2; r = s >> 4; o_p1 = use16 ? r : o_p1;
        s += p0 + q6 - p7 - p1; r = s >> 4; o_p0 = use16 ? r : o_p0;
        s += q0 + q7 - p7 - p0; r = s >> 4; o_q0 = use16 ? r : o_q0;
        s += q1 + q7 - p6 - q0; r = s >> 4; o_q1 = use16 ? r : o_q1;
        s += q2 + q7 - p5 - q1; r = s >> 4; o_q2 = use16 ? r : o_q2;
        s += q3 + q7 - p4 - q2; r = s >> 4; q[3] = use16 ? r : q3;
        s += q4 + q7 - p3 - q3; r = s >> 4; q[4] = use16 ? r : q4;
        s += q5 + q7 - p2 - q4; r = s >> 4; q[5] = use16 ? r : q5;
        s += q6 + q7 - p1 - q5; r = s >> 4; q[6] = use16 ? r : q6;
    }
    p[2] = o_p2; p[1] = o_p1; p[0] = o_p0; q[0] = o_q0; q[1] = o_q1; q[2] = o_q2;
}

/* Edge descriptor of one 8x8 block along the filtering direction: which filters run on its leading edge and on its inner
 * 4-sample edge, with the levels they use (built once per SB from the masks by svt_lf_desc_kernel).  The descriptor kernel turns
 * an entry into the two words the filter waves read: word 0 = thresholds of the leading edge (mblim | lim << 8 | hev_thr << 16) with
 * its width flags in the top byte, word 1 = thresholds of the inner edge with LF_KI in the top byte -- the filter loop then has no
 * dependent table look-up between reading a block's descriptor and filtering it. */
enum { LF_K16 = 1, LF_K8 = 2, LF_K4 = 4, LF_KI = 8 };
__device__ __forceinline__ uint32_t lf_entry(int flags, int lvl16, int lvl84, int lvli) {
    return (uint32_t)flags | ((uint32_t)lvl16 << 8) | ((uint32_t)lvl84 << 14) | ((uint32_t)lvli << 20);
}
__device__ __forceinline__ void lf_entry2(uint32_t e, const uint32_t *thr, uint32_t &w0, uint32_t &w1) {
    w0 = thr[((e & LF_K16) ? e >> 8 : e >> 14) & 63] | ((e & (LF_K16 | LF_K8 | LF_K4)) << 24);
    w1 = thr[(e >> 20) & 63] | ((e & LF_KI) << 24);
}

/* the filter of a block's leading edge between p and q ... */
__device__ __forceinline__ void lf_block_edges(int (&p)[8], int (&q)[8], uint32_t w0) {
    /* a position carries at most one width (eb_vp9_build_mask puts an edge into exactly one of the 16/8/4 masks) */
    const uint32_t fl = w0 >> 24;
    const int      kind = (fl & LF_K16) ? 16 : (fl & LF_K8) ? 8 : (fl & LF_K4) ? 4 : 0;
    if (__builtin_amdgcn_ballot_w64(kind != 0)) filter_regs<16>(p, q, kind, w0);
}
/* ... and of the inner edge inside q (q[3..0] | q[4..7]): only ever the 4-wide filter */
__device__ __forceinline__ void lf_inner_edge(int (&q)[8], uint32_t w1) {
    if (__builtin_amdgcn_ballot_w64((w1 >> 24) != 0)) {
        int ip[8] = {q[3], q[2], q[1], q[0], 0, 0, 0, 0}, iq[8] = {q[4], q[5], q[6], q[7], 0, 0, 0, 0};
        filter_regs<4>(ip, iq, (w1 >> 24) ? 4 : 0, w1);
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
/* pointers taken from the picture descriptors are global memory: global_load / global_store instead of flat accesses */
#define LF_GLOBAL __attribute__((address_space(1)))
#define LF_AS_GLOBAL(T, p) ((T LF_GLOBAL *)(uintptr_t)(p))
#define LF_LDS __attribute__((address_space(3)))

__device__ __forceinline__ uint32_t horiz_entry(int cb, unsigned m16, unsigned m8, unsigned m4, unsigned mi4, const uint8_t LF_GLOBAL *lfl, int lstep) {
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

/* raw fetch of a block's 8 samples (two dwords along a row, 8 bytes down a column) and its unpacking: split so that the fetch
 * of block c + 1 can be issued before block c is filtered (LDS round trips are ~100 cycles and this wave has nothing else to
 * hide them with) */
struct lf_raw { uint32_t w[8]; };
template <bool ROW> __device__ __forceinline__ void lf_fetch(const uint8_t *s, int st, lf_raw &r) {
    if (ROW) { const uint32_t *w = (const uint32_t *)s; r.w[0] = w[0]; r.w[1] = w[1]; }
    else { _Pragma("unroll") for (int i = 0; i < 8; i++) r.w[i] = s[i * st]; }
}
template <bool ROW> __device__ __forceinline__ void lf_unpack(const lf_raw &r, int (&v)[8]) {
    if (ROW) {
        const uint32_t a = r.w[0], b = r.w[1];
        v[0] = (int)(a & 0xff); v[1] = (int)((a >> 8) & 0xff); v[2] = (int)((a >> 16) & 0xff); v[3] = (int)(a >> 24);
        v[4] = (int)(b & 0xff); v[5] = (int)((b >> 8) & 0xff); v[6] = (int)((b >> 16) & 0xff); v[7] = (int)(b >> 24);
    } else { _Pragma("unroll") for (int i = 0; i < 8; i++) v[i] = (int)r.w[i]; }
}

/* One line of samples (a tile row for the vertical edges, a tile column for the horizontal ones) through its nblk
 * blocks: p slides along in registers, every sample is read and written once.  s = first sample of block 0 (the 8
 * samples before it are the neighbour's), tab[2 * c * tstep], tab[2 * c * tstep + 1] = descriptor words of this line's block c.
 * Samples and descriptor of block c + 1 are fetched while block c is filtered: nothing block c's filters write lies in block
 * c + 1 (the widest filter changes q0..q6 of its own block). */
template <bool ROW>
__device__ __forceinline__ void lf_line(uint8_t *s, int st, int nblk, const uint32_t *tab, int tstep) {
    int    p[8], q[8];
    lf_raw nxt;
    if (ROW) row_load8(s - 8, p, true); else col_load8(s - 8 * st, st, p, true);
    uint32_t w0 = tab[0], w1 = tab[1];
    lf_fetch<ROW>(s, st, nxt);
    for (int c = 0; c < nblk; c++) {
        lf_unpack<ROW>(nxt, q);
        const uint32_t e0 = w0, e1 = w1;
        if (c + 1 < nblk) { /* wave-uniform */
            w0 = tab[2 * (c + 1) * tstep]; w1 = tab[2 * (c + 1) * tstep + 1];
            lf_fetch<ROW>(ROW ? s + 8 * (c + 1) : s + 8 * (c + 1) * st, st, nxt);
        }
        lf_block_edges(p, q, e0);
        if (ROW) row_store8(s + 8 * c - 8, p, true); else col_store8(s + (8 * c - 8) * st, st, p, true);
        lf_inner_edge(q, e1);
        _Pragma("unroll") for (int i = 0; i < 8; i++) p[i] = q[7 - i];
    }
    if (ROW) row_store8(s + 8 * nblk - 8, p, true); else col_store8(s + (8 * nblk - 8) * st, st, p, true);
}

/* Copy a rectangle of 8-byte units between global memory and an LDS tile.  Tile sample (tx, ty) <-> plane sample
 * (x0 + tx, y0 + ty); tx in [tx0, tx0 + nx), ty in [ty0, ty0 + ny); x0 + tx0, nx and the LDS offsets are multiples of 8.
 * Rows with seam[ty] != 0 belong to a seam between two SB rows: they are written by one workgroup and read by the
 * workgroup of the next SB row inside the same launch, so they move with agent-scope (sc1, write-through / L2-bypass)
 * 8-byte accesses -- no release/acquire fence is needed for them (cdna_hip_programming.md, guideline 16).  Up to UNITS
 * loads per lane are in flight before the first store.  Plane rows must be 8-byte (wide) or 4-byte aligned. */
/* t / nu for the tile copies (units per tile row: at most 18): reciprocals from constant memory, one v_mul_hi per unit instead
 * of an integer division (~25 instructions) -- the copy waves spent most of their instructions dividing */
struct lf_magic_table {
    uint32_t v[33];
    constexpr lf_magic_table() : v() { for (uint32_t d = 1; d <= 32; d++) v[d] = (uint32_t)(0xffffffffu / d) + 1u; }
};
__constant__ const lf_magic_table lf_magics = lf_magic_table();
__device__ __forceinline__ int lf_udiv(int t, int d, uint32_t inv) { return d > 32 ? t / d : inv ? (int)__umulhi((uint32_t)t, inv) : t; }

template <int UNITS, typename UT>
__device__ __forceinline__ void tile_io_t(bool load, uint8_t *g, int gstride, uint8_t *l, int lstride, int x0, int y0, int tx0, int ty0,
                                          int nx, int ny, int seam_lo_end, int seam_hi_begin, int lane, int nlanes,
                                          const volatile int LF_LDS *wait_flag, int wait_val) {
    constexpr int UB = (int)sizeof(UT);      /* unit = 8 bytes, or 4 when the plane rows are only 4-byte aligned */
    const int nu = nx / UB, total = nu * ny;
    const uint32_t inv = lf_magics.v[nu <= 32 ? nu : 0];
    for (int t0 = lane; t0 < total; t0 += UNITS * nlanes) {
        UT  v[UNITS];
        int lo[UNITS];
        _Pragma("unroll") for (int u = 0; u < UNITS; u++) {
            const int t = t0 + u * nlanes;
            lo[u] = -1;
            if (t < total) {
                const int  r = lf_udiv(t, nu, inv), ty = ty0 + r, tx = tx0 + UB * (t - r * nu);
                UT LF_GLOBAL *gp = LF_AS_GLOBAL(UT, g + (ptrdiff_t)(y0 + ty) * gstride + x0 + tx);
                const bool seam = ty < seam_lo_end || ty >= seam_hi_begin;
                lo[u] = ty * lstride + tx;
                if (load) v[u] = seam ? __hip_atomic_load(gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *gp;
                else {
                    const uint32_t *lp = (const uint32_t *)(l + lo[u]);
                    UT              w = (UT)lp[0];
                    if constexpr (UB == 8) w |= (UT)lp[1] << 32;
                    if (seam) __hip_atomic_store(gp, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else *gp = w;
                }
            }
        }
        if (load) {
            /* the loads are in flight; the LDS buffer they land in may still be read by the write-back wave */
            if (wait_flag) while (*wait_flag < wait_val) __builtin_amdgcn_s_sleep(1);
            _Pragma("unroll") for (int u = 0; u < UNITS; u++)
                if (lo[u] >= 0) {
                    uint32_t *lp = (uint32_t *)(l + lo[u]);
                    lp[0] = (uint32_t)v[u];
                    if constexpr (UB == 8) lp[1] = (uint32_t)(v[u] >> 32);
                }
        }
    }
}
template <int UNITS>
__device__ __forceinline__ void tile_io(bool wide, bool load, uint8_t *g, int gstride, uint8_t *l, int lstride, int x0, int y0, int tx0,
                                        int ty0, int nx, int ny, int seam_lo_end, int seam_hi_begin, int lane, int nlanes,
                                        const volatile int LF_LDS *wait_flag = nullptr, int wait_val = 0) {
    if (wide) tile_io_t<UNITS, unsigned long long>(load, g, gstride, l, lstride, x0, y0, tx0, ty0, nx, ny, seam_lo_end, seam_hi_begin, lane, nlanes, wait_flag, wait_val);
    else tile_io_t<2 * UNITS, unsigned int>(load, g, gstride, l, lstride, x0, y0, tx0, ty0, nx, ny, seam_lo_end, seam_hi_begin, lane, nlanes, wait_flag, wait_val);
}

struct lf_geom { /* per (SB row, SB column): what exists of the tile */
    int x0, y0, cx0, cy0, hx, hy, vw, vh, cvw, cvh;
};
__device__ __forceinline__ lf_geom lf_geometry(int sb_row, int sc, int W, int H) {
    lf_geom g;
    const int CW = W >> 1, CH = H >> 1;
    g.x0 = sc * 64; g.y0 = sb_row * 64; g.cx0 = sc * 32; g.cy0 = sb_row * 32;
    g.hx = sc > 0 ? 8 : 0; g.hy = sb_row > 0 ? 8 : 0; /* halo of 8 up/left where there is a neighbour */
    g.vw = (W - g.x0) < 64 ? W - g.x0 : 64; g.vh = (H - g.y0) < 64 ? H - g.y0 : 64;
    g.cvw = (CW - g.cx0) < 32 ? CW - g.cx0 : 32; g.cvh = (CH - g.cy0) < 32 ? CH - g.cy0 : 32;
    return g;
}

/* Edge descriptors of every SB of every picture: no dependencies, one 128-thread workgroup per SB */
__global__ __launch_bounds__(128) void svt_lf_desc_kernel(const lf_pic_dev *__restrict__ pics, int rows_per_pic, int max_cols, svt_lf_thresh thr) {
    __shared__ uint32_t s_thr[64]; /* mblim | lim << 8 | hev_thr << 16 per filter level */
    if (threadIdx.x < 64) s_thr[threadIdx.x] = (uint32_t)thr.mblim[threadIdx.x] | ((uint32_t)thr.lim[threadIdx.x] << 8) | ((uint32_t)thr.hev_thr[threadIdx.x] << 16);
    __syncthreads();
    const int pic = blockIdx.x / (rows_per_pic * max_cols), rem = blockIdx.x - pic * rows_per_pic * max_cols;
    const int sb_row = rem / max_cols, sc = rem - sb_row * max_cols;
    const lf_pic_dev P = pics[pic];
    const int sb_cols = (P.mi_cols + 7) >> 3, sb_rows = (P.mi_rows + 7) >> 3;
    if (sb_row >= sb_rows || sc >= sb_cols) return;
    const int tid = threadIdx.x, mi_row = sb_row * 8;
    uint32_t LF_GLOBAL *d = LF_AS_GLOBAL(uint32_t, const_cast<uint32_t *>(P.desc) + ((size_t)sb_row * sb_cols + sc) * LF_DESC_WORDS);
    /* the masks are adjusted on a register copy; the per-8x8 levels are read straight from HBM (indexing them in the
     * copy would push the whole struct to scratch memory) */
    const svt_lf_mask LF_GLOBAL *gm = LF_AS_GLOBAL(const svt_lf_mask, &P.lfm[sb_row * P.lfm_stride + sc]);
    const uint8_t LF_GLOBAL     *lfl = gm->lfl_y;
    svt_lf_mask        m;
    _Pragma("unroll") for (int i = 0; i < 4; i++) { m.left_y[i] = gm->left_y[i]; m.above_y[i] = gm->above_y[i]; m.left_uv[i] = gm->left_uv[i]; m.above_uv[i] = gm->above_uv[i]; }
    m.int_4x4_y = gm->int_4x4_y; m.int_4x4_uv = gm->int_4x4_uv;
    adjust_mask(m, mi_row, sc * 8, P.mi_rows, P.mi_cols);
    if (tid < 64) {
        const int rr = tid >> 3, c = tid & 7, pair = rr >> 1, half = rr & 1;
        uint32_t w0, w1;
        lf_entry2(vert_entry(c, 8, (unsigned)(m.left_y[2] >> (16 * pair)) & 0xffff, (unsigned)(m.left_y[1] >> (16 * pair)) & 0xffff,
                            (unsigned)(m.left_y[0] >> (16 * pair)) & 0xffff, (unsigned)(m.int_4x4_y >> (16 * pair)) & 0xffff, half,
                            lfl[(2 * pair) * 8 + c], lfl[(2 * pair + 1) * 8 + c]), s_thr, w0, w1);
        d[2 * tid] = w0; d[2 * tid + 1] = w1;
        const int r = rr;
        unsigned  a16 = 0, a8 = 0, a4 = 0;
        if (mi_row + r != 0) { a16 = (unsigned)(m.above_y[2] >> (8 * r)) & 0xff; a8 = (unsigned)(m.above_y[1] >> (8 * r)) & 0xff; a4 = (unsigned)(m.above_y[0] >> (8 * r)) & 0xff; }
        lf_entry2(horiz_entry(c, a16, a8, a4, (unsigned)(m.int_4x4_y >> (8 * r)) & 0xff, &lfl[r * 8], 1), s_thr, w0, w1);
        d[128 + 2 * tid] = w0; d[129 + 2 * tid] = w1;
    } else if (tid < 80) {
        const int t = tid - 64, rr = t >> 2, c = t & 3, pair = rr >> 1, half = rr & 1;
        uint32_t w0, w1;
        lf_entry2(vert_entry(c, 4, (unsigned)(m.left_uv[2] >> (8 * pair)) & 0xff, (unsigned)(m.left_uv[1] >> (8 * pair)) & 0xff,
                                (unsigned)(m.left_uv[0] >> (8 * pair)) & 0xff, (unsigned)(m.int_4x4_uv >> (8 * pair)) & 0xff, half,
                                lfl[(4 * pair) * 8 + 2 * c], lfl[(4 * pair + 2) * 8 + 2 * c]), s_thr, w0, w1);
        d[256 + 2 * t] = w0; d[257 + 2 * t] = w1;
    } else if (tid < 96) {
        const int t = tid - 80, ru = t >> 2, c = t & 3, r = 2 * ru;
        unsigned  a16 = 0, a8 = 0, a4 = 0;
        if (mi_row + r != 0) { a16 = (unsigned)(m.above_uv[2] >> (4 * ru)) & 0xf; a8 = (unsigned)(m.above_uv[1] >> (4 * ru)) & 0xf; a4 = (unsigned)(m.above_uv[0] >> (4 * ru)) & 0xf; }
        const unsigned mi4 = (mi_row + r == P.mi_rows - 1) ? 0u : ((unsigned)(m.int_4x4_uv >> (4 * ru)) & 0xf);
        uint32_t w0, w1;
        lf_entry2(horiz_entry(c, a16, a8, a4, mi4, &lfl[r * 8], 2), s_thr, w0, w1);
        d[288 + 2 * t] = w0; d[289 + 2 * t] = w1;
    }
}

/* Workgroup = one SB row, four waves with fixed roles:
 *   wave 0  filters luma, wave 1 filters Cb (lanes 0-31) and Cr (lanes 32-63): vertical edges, then horizontal edges of the
 *           SB in tile buffer (sc & 1); the 8 rightmost columns (not final yet: the next SB's left edge filters modify
 *           them) are handed to the other buffer as the next SB's left halo -- every sample crosses HBM once per SB row;
 *   wave 2  stages the NEXT SB meanwhile: waits for the SB row above, loads the 64 new columns + top halo and the edge
 *           descriptors (svt_lf_desc_kernel) into the other buffer;
 *   wave 3  writes the finished columns of the tile filtered in the previous step back and publishes the row's progress.
 * One workgroup barrier per SB; wave 2 lets its loads fly while wave 3 is still reading the buffer they will land in.  Tile rows 0-7 (top halo) and the SB's last 8 rows are seam rows (sc1 accesses). */
template <bool early> /* seam rows are handed to the SB row below early, see step (c) (the default; <false>: with the tile's write-back) */
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void svt_lf_kernel(const lf_pic_dev *__restrict__ pics, int n_pics,
                                                     uint32_t *__restrict__ ticket, int rows_per_pic, int prof) {
    __shared__ __align__(16) uint8_t ytile[2][YROWS * YS];
    __shared__ __align__(16) uint8_t ctile[2][2][CROWS * CS];
    __shared__ uint32_t              s_desc[2][LF_DESC_WORDS]; /* edge descriptors, two words each: luma vertical [band][block] 64, luma
                                                                  horizontal 64, chroma vertical 16, chroma horizontal 16 */
    __shared__ int                   s_job;
    __shared__ int                   s_stored_;             /* last SB whose tile wave 2 has read back out of LDS */
    __shared__ int                   s_halo_;               /* last SB whose top halo rows (the SB row above's bottom rows) are in LDS */
    __shared__ int                   s_vdone_[2];           /* last SB whose vertical-edge pass the luma / the chroma wave has finished */
    const int tid = threadIdx.x;
    /* Persistent workgroups: each takes SB rows by ticket until none is left.  Tickets run over the rows of all pictures
     * interleaved (row 0 of every picture, then row 1, ...): the row a workgroup depends on, (pic, sb_row - 1), always
     * holds an earlier ticket and is therefore being worked on -- no residency assumption, no deadlock -- and only as
     * many rows are resident as the wavefront can keep busy, instead of every row of every picture spinning on its
     * predecessor while it occupies LDS and registers other kernels could use. */
    for (;;) {
    __syncthreads(); /* the previous row of this workgroup is complete (its last write-back included) */
    /* the two flags are polled: keep them explicit LDS references so the polls are ds_reads, not flat loads */
    volatile int LF_LDS &s_stored = *(volatile int LF_LDS *)&s_stored_;
    volatile int LF_LDS &s_halo   = *(volatile int LF_LDS *)&s_halo_;
    volatile int LF_LDS *s_vdone  = (volatile int LF_LDS *)s_vdone_;
    if (tid == 0) { s_job = (int)atomicAdd(ticket, 1u); s_stored = -1; s_halo = -1; s_vdone[0] = -1; s_vdone[1] = -1; }
    __syncthreads();
    const int job = s_job;
    if (job >= n_pics * rows_per_pic) break;
    const lf_pic_dev P = pics[job % n_pics];
    const int sb_row = job / n_pics;
    const int sb_cols = (P.mi_cols + 7) >> 3, sb_rows = (P.mi_rows + 7) >> 3;
    if (sb_row >= sb_rows) continue;
    const int W = P.planes.width, H = P.planes.height;
    const int mi_row = sb_row * 8;
    const int wave = tid >> 6, lane = tid & 63;
    const int nrows = P.mi_rows - mi_row < 8 ? P.mi_rows - mi_row : 8; /* 8-row bands of this SB row */
    const bool wide_y = (((uintptr_t)P.planes.y | (uintptr_t)P.planes.y_stride) & 7) == 0;
    const bool wide_c = (((uintptr_t)P.planes.u | (uintptr_t)P.planes.v | (uintptr_t)P.planes.uv_stride) & 7) == 0;
    unsigned long long tm_ = prof ? __builtin_amdgcn_s_memtime() : 0;
    const bool rowts = prof && job % n_pics == 0 && sb_row < 64;
#define LF_ROWTS(k, who, cond) do { if (rowts && tid == (who) && (cond)) g_lf_rowts[sb_row][k] = __builtin_amdgcn_s_memtime(); } while (0)
    LF_ROWTS(0, 0, true);
#define LF_MARK(i, who) do { if (prof && tid == (who)) { unsigned long long n_ = __builtin_amdgcn_s_memtime(); atomicAdd(&g_lf_prof[i], n_ - tm_); tm_ = n_; } } while (0)

    /* wave 2 stages SB `sc` into buffer sc & 1 in two parts.  stage(sc): the SB's own rows and its edge descriptors --
     * nothing of it depends on another SB row, it runs one SB ahead of the filter waves.  stage_halo(sc): the 8 rows
     * above the SB, which the SB row above finishes with (sb_row-1, sc+1); only the horizontal-edge pass needs them, so
     * they are fetched while the vertical-edge pass of the same SB runs: the hand-over between SB rows has the poll, 8
     * rows and one pass on its critical path instead of a whole tile and both passes.  The left halo comes from the
     * filter waves, except for SB 0 which has none. */
    auto stage_halo = [&](int sc) {
        const int     buf = sc & 1;
        const lf_geom g = lf_geometry(sb_row, sc, W, H);
        if (sb_row > 0) { /* wait for the SB row above to have handed over everything above this SB: its tile sc (columns up to 56)
                             and, early, the bottom rows of its last 8 columns, which (sb_row-1, sc+1)'s vertical pass finishes */
            const uint32_t need = early ? (uint32_t)(sc + 1) : (uint32_t)(sc + 2 < sb_cols ? sc + 2 : sb_cols);
            if (lane == 0)
                while (__hip_atomic_load(LF_AS_GLOBAL(uint32_t, &P.progress[sb_row - 1]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) __builtin_amdgcn_s_sleep(1);
            LF_MARK(0, 128);
            tile_io<2>(wide_y, true, P.planes.y, P.planes.y_stride, ytile[buf], YS, g.x0 - 8, g.y0 - 8, 8, 0, g.vw, 8, 8, 8 + g.vh - 8, lane, 64);
            if (!P.y_only) {
                tile_io<1>(wide_c, true, P.planes.u, P.planes.uv_stride, ctile[buf][0], CS, g.cx0 - 8, g.cy0 - 8, 8, 0, g.cvw, 8, 8, 8 + g.cvh - 8, lane, 64);
                tile_io<1>(wide_c, true, P.planes.v, P.planes.uv_stride, ctile[buf][1], CS, g.cx0 - 8, g.cy0 - 8, 8, 0, g.cvw, 8, 8, 8 + g.cvh - 8, lane, 64);
            }
        }
        /* LDS accesses of one wave are ordered: the flag lands after the rows */
        if (lane == 0) s_halo = sc;
        LF_ROWTS(1, 128, sc == 0);
    };
    auto stage = [&](int sc) {
        const int     buf = sc & 1;
        const lf_geom g = lf_geometry(sb_row, sc, W, H);
        /* tile columns 8.. hold the SB's own samples, rows 8..; seam rows: the SB's last 8 rows */
        tile_io<11>(wide_y, true, P.planes.y, P.planes.y_stride, ytile[buf], YS, g.x0 - 8, g.y0 - 8, 8, 8, g.vw, g.vh, 8, 8 + g.vh - 8, lane, 64, &s_stored, sc - 2);
        if (!P.y_only) {
            tile_io<3>(wide_c, true, P.planes.u, P.planes.uv_stride, ctile[buf][0], CS, g.cx0 - 8, g.cy0 - 8, 8, 8, g.cvw, g.cvh, 8, 8 + g.cvh - 8, lane, 64);
            tile_io<3>(wide_c, true, P.planes.v, P.planes.uv_stride, ctile[buf][1], CS, g.cx0 - 8, g.cy0 - 8, 8, 8, g.cvw, g.cvh, 8, 8 + g.cvh - 8, lane, 64);
        }
        LF_MARK(1, 128);
        /* edge descriptors of this SB (built by svt_lf_desc_kernel): luma vertical 64, luma horizontal 64, chroma 16 + 16 */
        {
            const uint32_t LF_GLOBAL *d = LF_AS_GLOBAL(const uint32_t, P.desc + ((size_t)sb_row * sb_cols + sc) * LF_DESC_WORDS);
            _Pragma("unroll") for (int k = 0; k < LF_DESC_WORDS / 64; k++) s_desc[buf][k * 64 + lane] = d[k * 64 + lane];
        }
        LF_MARK(2, 128);
    };

    if (wave == 2) stage(0);
    __syncthreads();
    for (int sc = 0; sc < sb_cols; sc++) {
        const int     buf = sc & 1;
        const lf_geom g = lf_geometry(sb_row, sc, W, H);
        const bool    last = sc + 1 == sb_cols;
        if (wave == 2) {
            stage_halo(sc);
            if (!last) stage(sc + 1);
        } else if (wave == 0) {
            /* vertical edges: lane = sample row; horizontal edges: lane = sample column (LDS accesses of one wave are
             * ordered, no barrier between the two passes) */
            if (lane < g.vh) lf_line<true>(ytile[buf] + (8 + lane) * YS + 8, 1, 8, &s_desc[buf][2 * (lane >> 3) * 8], 1);
            if (early && lane == 0) s_vdone[0] = sc; /* LDS accesses of one wave are ordered: the flag lands behind the pass's stores */
            LF_ROWTS(4, 0, sc == 1);
            LF_MARK(3, 0);
            while (s_halo < sc) __builtin_amdgcn_s_sleep(1); /* the rows above the SB have arrived */
            if (lane < g.vw) lf_line<false>(ytile[buf] + 8 * YS + 8 + lane, YS, nrows, &s_desc[buf][128 + 2 * (lane >> 3)], 8);
            LF_MARK(4, 0);
            LF_ROWTS(2, 0, sc == 0);
            LF_ROWTS(5, 0, last);
            if (!last) {
                while (s_stored < sc - 1) __builtin_amdgcn_s_sleep(1); /* the other buffer has been written back */
                for (int r = lane; r < YROWS; r += 64) {
                    const uint32_t *src = (const uint32_t *)(ytile[buf] + r * YS + 64);
                    uint32_t       *dst = (uint32_t *)(ytile[buf ^ 1] + r * YS);
                    dst[0] = src[0]; dst[1] = src[1];
                }
            }
        } else if (wave == 1 && !P.y_only) {
            const int pl = lane >> 5, l5 = lane & 31;
            if (l5 < g.cvh) lf_line<true>(ctile[buf][pl] + (8 + l5) * CS + 8, 1, 4, &s_desc[buf][256 + 2 * (l5 >> 3) * 4], 1);
            if (early && lane == 0) s_vdone[1] = sc;
            while (s_halo < sc) __builtin_amdgcn_s_sleep(1);
            if (l5 < g.cvw) lf_line<false>(ctile[buf][pl] + 8 * CS + 8 + l5, CS, (nrows + 1) >> 1, &s_desc[buf][288 + 2 * (l5 >> 3)], 4);
            if (!last) {
                while (s_stored < sc - 1) __builtin_amdgcn_s_sleep(1);
                for (int r = l5; r < CROWS; r += 32) {
                    const uint32_t *src = (const uint32_t *)(ctile[buf][pl] + r * CS + 32);
                    uint32_t       *dst = (uint32_t *)(ctile[buf ^ 1][pl] + r * CS);
                    dst[0] = src[0]; dst[1] = src[1];
                }
            }
        }
        __syncthreads();
        LF_MARK(5, 0);
        if (wave == 3) {
            /* Write back what is final.  (a) the SB's own columns 8 .. 8+56 (.. 8+vw for the last SB), rows 8-hy .. 8+vh; (b) the 8
             * left-halo columns (the previous SB's last columns, which this SB's vertical pass finished) -- without their bottom 8
             * rows when an SB row below waits for them: those went out early, in the previous iteration's step (c), and the row
             * below may already have filtered and rewritten them. */
            const bool below = early && sb_row + 1 < sb_rows;
            if (!early) { /* throughput mode: one rectangle per plane (tile columns 8-hx .. 8+56, rows 8-hy .. 8+vh), nothing handed over early */
                const int nxs = (last ? g.vw : 56) + g.hx, nxc = (last ? g.cvw : 24) + g.hx;
                tile_io<11>(wide_y, false, P.planes.y, P.planes.y_stride, ytile[buf], YS, g.x0 - 8, g.y0 - 8, 8 - g.hx, 8 - g.hy, nxs, g.vh + g.hy, 8, 8 + g.vh - 8, lane, 64);
                if (!P.y_only) {
                    tile_io<3>(wide_c, false, P.planes.u, P.planes.uv_stride, ctile[buf][0], CS, g.cx0 - 8, g.cy0 - 8, 8 - g.hx, 8 - g.hy, nxc, g.cvh + g.hy, 8, 8 + g.cvh - 8, lane, 64);
                    tile_io<3>(wide_c, false, P.planes.v, P.planes.uv_stride, ctile[buf][1], CS, g.cx0 - 8, g.cy0 - 8, 8 - g.hx, 8 - g.hy, nxc, g.cvh + g.hy, 8, 8 + g.cvh - 8, lane, 64);
                }
            } else {
            const int  nown = last ? g.vw : 56, nownc = last ? g.cvw : 24, cut = below ? 8 : 0;
            tile_io<11>(wide_y, false, P.planes.y, P.planes.y_stride, ytile[buf], YS, g.x0 - 8, g.y0 - 8, 8, 8 - g.hy, nown, g.vh + g.hy, 8, 8 + g.vh - 8, lane, 64);
            if (g.hx) tile_io<2>(wide_y, false, P.planes.y, P.planes.y_stride, ytile[buf], YS, g.x0 - 8, g.y0 - 8, 0, 8 - g.hy, 8, g.vh + g.hy - cut, 8, 8 + g.vh - 8, lane, 64);
            if (!P.y_only) {
                _Pragma("unroll") for (int pl = 0; pl < 2; pl++) {
                    uint8_t *cp = pl ? P.planes.v : P.planes.u;
                    tile_io<3>(wide_c, false, cp, P.planes.uv_stride, ctile[buf][pl], CS, g.cx0 - 8, g.cy0 - 8, 8, 8 - g.hy, nownc, g.cvh + g.hy, 8, 8 + g.cvh - 8, lane, 64);
                    if (g.hx) tile_io<1>(wide_c, false, cp, P.planes.uv_stride, ctile[buf][pl], CS, g.cx0 - 8, g.cy0 - 8, 0, 8 - g.hy, 8, g.cvh + g.hy - cut, 8, 8 + g.cvh - 8, lane, 64);
                }
            }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); /* the tile has left LDS: the buffer may be refilled */
            if (lane == 0) s_stored = sc;
            LF_MARK(6, 192);
            /* (c) early hand-over to the SB row below: the bottom 8 rows of this SB's LAST 8 columns are final as soon as the NEXT SB's
             * vertical pass (running right now in the other buffer, whose left-halo columns they are) has filtered its left edge --
             * its horizontal pass never touches the halo columns.  Handing them over now instead of with the next tile's
             * write-back lets row r + 1 follow 1.2 instead of 2 SB steps behind row r: 1.09 -> 0.89 ms for one 4K picture, 1.21 -> 1.03 ms
             * for 16 (every row with a workgroup of its own, see the launcher). */
            if (!last && below) {
                while (s_vdone[0] < sc + 1) __builtin_amdgcn_s_sleep(1);
                const lf_geom gn = lf_geometry(sb_row, sc + 1, W, H);
                tile_io<1>(wide_y, false, P.planes.y, P.planes.y_stride, ytile[buf ^ 1], YS, gn.x0 - 8, gn.y0 - 8, 0, 8 + gn.vh - 8, 8, 8, 1 << 20, 0, lane, 64);
                if (!P.y_only) {
                    while (s_vdone[1] < sc + 1) __builtin_amdgcn_s_sleep(1);
                    tile_io<1>(wide_c, false, P.planes.u, P.planes.uv_stride, ctile[buf ^ 1][0], CS, gn.cx0 - 8, gn.cy0 - 8, 0, 8 + gn.cvh - 8, 8, 8, 1 << 20, 0, lane, 64);
                    tile_io<1>(wide_c, false, P.planes.v, P.planes.uv_stride, ctile[buf ^ 1][1], CS, gn.cx0 - 8, gn.cy0 - 8, 0, 8 + gn.cvh - 8, 8, 8, 1 << 20, 0, lane, 64);
                }
            }
            /* publish: every store of this wave has completed (seam rows were written through) -> progress counter.  sc + 1 = "everything
             * above SB sc of the row below is in memory" */
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_store(LF_AS_GLOBAL(uint32_t, &P.progress[sb_row]), (uint32_t)(sc + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            LF_ROWTS(3, 192, sc == 0);
            LF_ROWTS(6, 192, last);
            LF_MARK(7, 192);
        }
    }
#undef LF_MARK
#undef LF_ROWTS
    } /* next ticket */
}
} // namespace

static int32_t lf_launch(svt_hip_ctx *ctx, int n_pics, const svt_yuv_planes *d_recon, const svt_lf_mask *const *d_lfm, const int32_t *lfm_stride,
                         const svt_lf_thresh *thr, const int32_t *mi_rows, const int32_t *mi_cols, int32_t y_only) {
    int max_rows = 0, max_cols = 0;
    for (int i = 0; i < n_pics; i++) {
        if (mi_rows[i] < 1 || mi_cols[i] < 1 || d_recon[i].width != mi_cols[i] * 8 || d_recon[i].height != mi_rows[i] * 8)
            return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "lf: plane size must be mi_cols*8 x mi_rows*8");
        const svt_yuv_planes &pl = d_recon[i];
        if ((((uintptr_t)pl.y | (uintptr_t)pl.y_stride) & 3) || (!y_only && (((uintptr_t)pl.u | (uintptr_t)pl.v | (uintptr_t)pl.uv_stride) & 3)))
            return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "lf: plane pointers and strides must be multiples of 4 bytes");
        const int r = (mi_rows[i] + 7) / 8, cc = (mi_cols[i] + 7) / 8;
        if (r > max_rows) max_rows = r;
        if (cc > max_cols) max_cols = cc;
    }
    HIP_TRY(hipSetDevice(ctx->device));
    /* descriptors + progress counters + ticket in one device scratch block */
    const size_t desc_bytes = (sizeof(lf_pic_dev) * (size_t)n_pics + 15) & ~(size_t)15;
    const size_t cnt_words  = (size_t)n_pics * max_rows + 4;
    lf_pic_dev  *h = nullptr;
    uint8_t     *d = nullptr;
    if (svt_ctx_stage(ctx, desc_bytes + cnt_words * 4, (void **)&h, (void **)&d)) return svt_set_error(SVT_HIP_ERR_NO_RESOURCES, "lf: scratch");
    uint32_t *cnt = (uint32_t *)(d + desc_bytes);
    /* per-SB edge descriptors of all pictures (grow-only context buffer) */
    uint32_t *edesc = (uint32_t *)svt_ctx_slot(ctx, 24, (size_t)n_pics * max_rows * max_cols * LF_DESC_WORDS * sizeof(uint32_t));
    if (!edesc) return svt_set_error(SVT_HIP_ERR_NO_RESOURCES, "lf: descriptor buffer");
    for (int i = 0; i < n_pics; i++) {
        h[i].planes = d_recon[i]; h[i].lfm = d_lfm[i]; h[i].lfm_stride = lfm_stride[i]; h[i].mi_rows = mi_rows[i]; h[i].mi_cols = mi_cols[i];
        h[i].y_only = y_only; h[i].progress = cnt + 4 + (size_t)i * max_rows;
        h[i].desc = edesc + (size_t)i * max_rows * max_cols * LF_DESC_WORDS;
    }
    HIP_TRY(hipMemcpyAsync(d, h, sizeof(lf_pic_dev) * (size_t)n_pics, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemsetAsync(cnt, 0, cnt_words * 4, ctx->stream));
    HIP_TRY(hipEventRecord(ctx->ev_start, ctx->stream));
    static const bool want_prof = getenv("SVT_HIP_LF_PROFILE") != nullptr;
    if (want_prof) { unsigned long long z[8] = {0}; HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_lf_prof), z, sizeof z)); }
    hipLaunchKernelGGL(svt_lf_desc_kernel, dim3(n_pics * max_rows * max_cols), dim3(128), 0, ctx->stream, (const lf_pic_dev *)d, max_rows, max_cols, *thr);
    /* Every SB row gets a workgroup (tickets: a row's predecessor always holds an earlier one) and the seam rows are handed to the row
     * below early (step (c) of the kernel).  Measured on MI355X, 2160p, launch alone: 1 picture 0.89 ms, 4 pictures 0.91 ms, 16 pictures
     * 1.03 ms.  Until round 5 launches of more than 4 pictures ran 16 rows per picture without the early hand-over ("only as many rows as
     * the wavefront keeps busy"): 1.45 ms for 16 pictures -- the row-to-row lag (two SB steps of ~6.8 us) times 34 rows IS most of a
     * picture's time, and a workgroup that waits for its predecessor costs the launch nothing while one that has not started cannot take
     * its seam rows the moment they appear.  Inside the pipelined step the change is neutral (other kernels fill the device either way);
     * one GOP at a time gains 6 %.  SVT_HIP_LF_ROWS / SVT_HIP_LF_EARLY override (experiments). */
    static const int rows_env = getenv("SVT_HIP_LF_ROWS") ? atoi(getenv("SVT_HIP_LF_ROWS")) : 0;
    const int rows_in_flight = rows_env > 0 ? rows_env : max_rows;
    const int lf_wgs = n_pics * (max_rows < rows_in_flight ? max_rows : rows_in_flight);
    static const int early_env = getenv("SVT_HIP_LF_EARLY") ? atoi(getenv("SVT_HIP_LF_EARLY")) : -1;
    const int early = early_env >= 0 ? early_env : 1;
    if (early) hipLaunchKernelGGL(svt_lf_kernel<true>, dim3(lf_wgs), dim3(256), 0, ctx->stream, (const lf_pic_dev *)d, n_pics, cnt, max_rows, want_prof ? 1 : 0);
    else hipLaunchKernelGGL(svt_lf_kernel<false>, dim3(lf_wgs), dim3(256), 0, ctx->stream, (const lf_pic_dev *)d, n_pics, cnt, max_rows, want_prof ? 1 : 0);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(ctx->ev_stop, ctx->stream));
    if (want_prof) {
        unsigned long long hp[8];
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        HIP_TRY(hipMemcpyFromSymbol(hp, HIP_SYMBOL(g_lf_prof), sizeof hp));
        static const char *nm[8] = {"io:wait", "io:tile_load", "io:descriptors", "filter:vertical", "filter:horizontal+halo", "filter:barrier",
                                    "io:tile_store", "io:publish"};
        unsigned long long tot = 0, steps = 0;
        for (int i = 3; i < 6; i++) tot += hp[i]; /* critical path = the luma filter wave */
        for (int i = 0; i < n_pics; i++) steps += (unsigned long long)((mi_rows[i] + 7) / 8) * ((mi_cols[i] + 7) / 8);
        fprintf(stderr, "[lf-profile] pics=%d SB steps=%llu avg cycles/SB=%llu :", n_pics, steps, tot / (steps ? steps : 1));
        for (int i = 0; i < 8; i++) fprintf(stderr, " %s=%.1f%%", nm[i], 100.0 * (double)hp[i] / (double)tot);
        fprintf(stderr, "\n");
        if (getenv("SVT_HIP_LF_ROWTS")) { /* the wavefront's hand-over between SB rows, picture 0 (ticks of s_memtime: 100 MHz) */
            static unsigned long long ts[64][8];
            HIP_TRY(hipMemcpyFromSymbol(ts, HIP_SYMBOL(g_lf_rowts), sizeof ts));
            for (int r = 0; r < max_rows && r < 64; r++) {
                fprintf(stderr, "[lf-rows] row %2d:", r);
                for (int k = 0; k < 7; k++) fprintf(stderr, " %9.2f", (double)(ts[r][k] - ts[0][0]) / 100.0);
                fprintf(stderr, "  (us: taken, halo 0, SB 0 done, publish 1, V(SB 1) done, last SB done, row complete)\n");
            }
        }
    }
    svt_ctx_stage_commit(ctx);
    ctx->timed = 1;
    return SVT_HIP_OK;
}

/* Takes the context's edge-descriptor buffer for launches of up to n_pics pictures of mi_rows x mi_cols now.  The buffer grows on demand, but
 * growing it waits for the context's stream -- behind the public API that was the first key frame's whole intra pass (4 ms) on the thread that
 * sent the picture, and a stall of the first deblocking launches of every new batch size. */
extern "C" int32_t svt_hip_lf_reserve(svt_hip_ctx *ctx, int32_t n_pics, int32_t mi_rows, int32_t mi_cols) {
    if (!ctx || n_pics < 1 || mi_rows < 1 || mi_cols < 1) return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "lf_reserve: bad argument");
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t rows = (size_t)(mi_rows + 7) / 8, cols = (size_t)(mi_cols + 7) / 8;
    if (!svt_ctx_slot(ctx, 24, (size_t)n_pics * rows * cols * LF_DESC_WORDS * sizeof(uint32_t))) return svt_set_error(SVT_HIP_ERR_NO_RESOURCES, "lf: descriptor buffer");
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
