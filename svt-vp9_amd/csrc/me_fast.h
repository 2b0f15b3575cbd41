/*
 * me_fast.h -- driver and phases of the ME kernel for the presets whose hierarchical search is ONE region at the
 * 1/16-resolution level only, with the row-subsampled SAD refinement of the 32x32 / 16x16 PUs (2160p, enc-mode 8 and up:
 * me_spec.h SPEC 1 -- BASELINE C3 / C4).  Same per-SB flow as me_sb_run (me_core.h), i.e. the reference's
 * motion_estimate_sb (Codec/EbMotionEstimation.c:4524-5305): test_search_area_bounds (:4260) -> hme_level0 on a single
 * quadrant (:2872-2920) -> check_zero_zero_center (:3758) -> full_pel_search_sb (:951) -> su_pel_enable (:3839) ->
 * interpolate_search_region_avc (:992) -> half / quarter-pel refinement (:1565, :2471) -> bi_prediction_search (:3695).
 *
 * What differs from the general driver is where the UNIFORM work runs.  The general driver plans each HME level on one
 * thread into an LDS work list, copies the reference's plane descriptors to LDS and reduces every sum through LDS atomics;
 * all of it is wave-uniform, and with the kernel's scalar registers exhausted it ran as vector instructions on spilled
 * scalars (v_readlane / v_writelane), four times per SB.  Here every wave derives the uniform geometry itself on the scalar
 * unit from the kernel arguments and the picture descriptor (scalar loads), per phase, from a handful of carried values
 * (search centre, area origin) -- nothing uniform is kept across phases that can be recomputed, so nothing spills -- and the
 * phases are written for exactly this flow:
 *   - centre tests: one 8-byte piece per thread and candidate, five sums packed two per dword through the DPP row
 *     reduction, ONE pair of 64-bit LDS adds per wave;
 *   - HME: no work list; window geometry in scalar registers, tasks by reciprocal multiplication, 32-bit keys, one
 *     32-bit LDS min per wave; the window stride of the common 64-wide area is a compile-time constant (immediate offsets);
 *   - su_pel_enable: the 4 + 16 motion vectors / SADs sit in two DPP rows of wave 0; the decision is taken there and
 *     published as two bits.
 * Everything else (full-pel, interpolation, refinement, output) is shared with me_core.h.  Device only: the CPU emulation
 * (tests/emu) runs the general driver, and the -m gpu parity tests against the oracle pin this one.
 */
#ifndef SVT_ME_FAST_H
#define SVT_ME_FAST_H
#include "me_core.h"

#ifndef SVT_HOST_EMU

/* the presets this driver serves (compile-time property of a specialised instance) */
constexpr bool me_fast_params_ok(uint8_t hme, uint8_t l0, uint8_t l1, uint8_t l2, uint8_t single, uint8_t method, uint8_t model, uint8_t f64,
                                 uint8_t cu16, uint8_t cu8) {
    return hme && l0 && !l1 && !l2 && single && method == SVT_SUB_SAD_SEARCH && model == 1 && !f64 && cu16 == 0 && cu8 == 1;
}

/* sums of the centre tests: slot s = 2 * list + (0: test_search_area_bounds, 1: check_zero_zero_center), five dwords each; they
 * live in the bytes of hme_sad (the general driver's per-level results: unused here) */
#if defined(ME_ASM_MARKS) /* static instruction counts (tools/me_static_counts.py): a comment in the assembly at every mark */
#define FME_MARK(i) __asm__ volatile("; @MARK %0" ::"n"(i))
#elif defined(ME_FINE_PROF) /* dynamic counts and times per phase (tools/me_phase_profile.sh): the kernel ends at mark g_me_stop_after of list 0 */
#define FME_MARK(i) do { if (g_me_stop_after == (i)) return; } while (0)
#else
#define FME_MARK(i) ((void)0)
#endif
#define FME_ACC(st, s) ((uint32_t *)(st)->hme_sad + 5 * (s))
/* HME arg-min key (sad << 16 | y << 8 | x) of list l: in the bytes of hme_keys */
#define FME_HKEY(st, l) ((uint32_t *)(st)->hme_keys + (l))
/* su_pel_enable decision of the current list: bit 0 = 32x32, bit 1 = 16x16 */
#define FME_GATE(st) (&(st)->supel[0])
/* Source planes of the sub-pel candidates as byte offsets, built once per SB (fph_init) in the bytes of spu (the general
 * driver's refined-PU list: unused here).  QTAB[method * 8 + position]: the two planes a quarter-pel candidate averages
 * (set_quarter_pel_refinement_inputs_on_the_fly :2290-2465); BTAB[frac]: the plane(s) of a prediction at a quarter-pel vector
 * (select_buffer :3310 / quarter_pel_compensation :3358; one plane: twice the same).  A source is 16 bits: bit 15 = the integer
 * plane (search region), bits 0-14 = 128 + byte offset from the lane's position in that class of planes (B / H / J share a
 * stride and lie plane_bytes apart). */
#define FME_QTAB(st) ((uint32_t *)(st)->spu)
#define FME_BTAB(st) ((uint32_t *)(st)->spu + 32)
SVT_DEV uint32_t fme_src_code(const me_ctx_t *c, int plane, int dx, int dy) {
    return plane == ME_PF ? (0x8000u | (uint32_t)(128 + ME_MUL(dy, c->L.region_stride) + dx))
                          : (uint32_t)(128 + ME_MUL(plane - 1, c->L.plane_bytes) + ME_MUL(dy, c->L.plane_stride) + dx);
}

/* inclusive sums over the wave: total in lane 63 (row_shr 1, 2, 4, 8, then row_bcast:15 into rows 1 / 3 and row_bcast:31 into
 * rows 2 / 3) */
SVT_DEV uint32_t fme_wave_sum63(uint32_t v) {
    v = svt_row_prefix_add(v, 16);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
    return v;
}
/* packed pair of 16-bit sums (each lane's halves <= 2040): lane 31 = lanes 0-31, lane 63 = lanes 32-63, no carry between the halves */
SVT_DEV uint32_t fme_half_sums_pk(uint32_t v) {
    v = svt_row_prefix_add(v, 16);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
    return v;
}

/* initial state, the 64x64 source SB and rows 0, 2, .. of its 1/16-resolution copy (Codec/EbMotionEstimationProcess.c:984-1035) */
SVT_DEV void fph_init(const me_ctx_t *c, int tid, int do_hme) {
    me_state_t *st = c->st;
    if (tid < 85) { st->best_mv[0][tid] = 0; st->best_mv[1][tid] = 0; st->best_sad[0][tid] = 0; st->best_sad[1][tid] = 0; }
    if (tid >= 96 && tid < 118) ((uint32_t *)st->dir)[tid - 96] = 0;
    if (tid >= 160 && tid < 162) FME_HKEY(st, tid - 160)[0] = 0xffffffffu;
    if (tid >= 192 && tid < 224) {
        const uint32_t e = me_qtab_get((tid - 192) >> 3, (tid - 192) & 7);
        FME_QTAB(st)[tid - 192] = fme_src_code(c, (int)(e & 3), -(int)((e >> 2) & 1), -(int)((e >> 3) & 1)) |
                                  (fme_src_code(c, (int)((e >> 4) & 3), -(int)((e >> 6) & 1), -(int)((e >> 7) & 1)) << 16);
    }
    if (tid >= 224 && tid < 240) {
        int            has_b;
        const uint32_t e = me_btab_get(tid - 224, &has_b), a = fme_src_code(c, (int)(e & 3), (int)((e >> 2) & 1), (int)((e >> 3) & 1));
        FME_BTAB(st)[tid - 224] = a | ((has_b ? fme_src_code(c, (int)((e >> 4) & 3), (int)((e >> 6) & 1), (int)((e >> 7) & 1)) : a) << 16);
    }
    const svt_plane *cf = &c->pic->cur.full;
    {
        const uint8_t *g = me_pix(cf, c->sb_x, c->sb_y);
        const int      row = tid >> 2, seg = tid & 3;
        const me_u32x4 v = me_ld128u_g(g + (uint32_t)(ME_MUL(row, cf->stride) + 16 * seg));
        uint32_t      *d = (uint32_t *)(c->src + row * ME_SB + 16 * seg);
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    if (do_hme && tid < 8) { /* whole SBs only: 16 samples per row */
        const svt_plane *cs = &c->pic->cur.sixteenth;
        const me_u32x4   v = me_ld128u_g(me_pix(cs, c->sb_x >> 2, c->sb_y >> 2) + (uint32_t)ME_MUL(2 * tid, cs->stride));
        uint32_t        *d = (uint32_t *)(st->sixteenth_sb + 16 * tid);
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
}

/* Row-subsampled 64-wide SADs of the source SB against up to five displaced reference blocks in global memory
 * (test_search_area_bounds :4260 / check_zero_zero_center :3758), by WAVE 0 ALONE: 5 x 2048 samples are too little work to pay
 * four waves' address set-up, reductions and a cross-wave meeting in LDS -- the kernel is bound by the sum of vector
 * instructions over its waves, and the other three waves wait at the barrier for the same global round trip either way.
 * Lane = half a row (32 bytes: two 16-byte loads per candidate); base + off[k] = top-left sample of candidate k.  The sums go to
 * slot[0..4]. */
SVT_DEV void fph_center(const me_ctx_t *c, int tid, const uint8_t *base, int rstride, int nc, const int32_t *off, uint32_t *slot) {
    if (ME_UNI(tid >> 6) != 0) return;
    const int r = tid >> 1, h = tid & 1;
    uint32_t  acc[5] = {0, 0, 0, 0, 0};
    if (2 * r < c->sb_h) {
        const uint32_t voff = (uint32_t)(ME_MUL(2 * r, rstride) + 32 * h);
        me_u32x4       v[5][2];
        _Pragma("unroll") for (int k = 0; k < 5; k++)
            if (k < nc) { v[k][0] = me_ld128u_g(base + off[k] + voff); v[k][1] = me_ld128u_g(base + off[k] + voff + 16); }
        const uint32_t *sp = (const uint32_t *)(c->src + (2 * r) * ME_SB + 32 * h);
        uint32_t        sv[8];
        _Pragma("unroll") for (int i = 0; i < 8; i++) sv[i] = sp[i];
        _Pragma("unroll") for (int k = 0; k < 5; k++)
            if (k < nc) {
                uint32_t t = svt_sad4(v[k][0].x, sv[0], 0);
                t = svt_sad4(v[k][0].y, sv[1], t); t = svt_sad4(v[k][0].z, sv[2], t); t = svt_sad4(v[k][0].w, sv[3], t);
                t = svt_sad4(v[k][1].x, sv[4], t); t = svt_sad4(v[k][1].y, sv[5], t); t = svt_sad4(v[k][1].z, sv[6], t);
                acc[k] = svt_sad4(v[k][1].w, sv[7], t);
            }
    }
    /* a lane's sums stay below 2^13: two candidates per dword through the first three steps (8 lanes < 2^16), then singly */
    uint32_t p[3] = {acc[0] | (acc[1] << 16), acc[2] | (acc[3] << 16), acc[4]};
    _Pragma("unroll") for (int i = 0; i < 3; i++)
        if (i < 2 || nc > 4) { p[i] = SVT_DPP_ADD(p[i], 0x111); p[i] = SVT_DPP_ADD(p[i], 0x112); p[i] = SVT_DPP_ADD(p[i], 0x114); }
    uint32_t a[5] = {p[0] & 0xffffu, p[0] >> 16, p[1] & 0xffffu, p[1] >> 16, p[2]};
    _Pragma("unroll") for (int k = 0; k < 5; k++)
        if (k < nc) {
            a[k] = SVT_DPP_ADD(a[k], 0x118);
            a[k] += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a[k], 0x142, 0xa, 0xf, false);
            a[k] += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a[k], 0x143, 0xc, 0xf, false);
        }
    if (tid == 63) { _Pragma("unroll") for (int k = 0; k < 5; k++) if (k < nc) slot[k] = a[k]; }
}
/* the sums back as scalars (after the barrier that follows fph_center) */
SVT_DEV void fme_center_sums(const uint32_t *slot, int nc, uint32_t s[5]) {
    _Pragma("unroll") for (int k = 0; k < 5; k++) s[k] = k < nc ? (uint32_t)ME_UNI(slot[k]) : 0u;
}

/* the same SAD for ONE displaced block that already sits in the LDS search region (region byte (col, row) = its top-left
 * sample), by wave 0: slot[0] */
SVT_DEV void fph_region_center(const me_ctx_t *c, int tid, int col, int row, uint32_t *slot) {
    if (ME_UNI(tid >> 6) != 0) return;
    const int r = tid >> 1, h = tid & 1, rs = c->L.region_stride;
    uint32_t  acc = 0;
    if (2 * r < c->sb_h) {
        const uint32_t  a = (uint32_t)(c->region - c->lds) + (uint32_t)(ME_MUL(row + 2 * r, rs) + col + 32 * h), sh = a & 3u;
        const uint32_t *q = (const uint32_t *)(c->lds + (a - sh));
        const uint32_t *sp = (const uint32_t *)(c->src + (2 * r) * ME_SB + 32 * h);
        uint32_t        lo = q[0];
        _Pragma("unroll") for (int i = 0; i < 8; i++) {
            const uint32_t hi = q[i + 1];
            acc = svt_sad4(svt_alignbyte(hi, lo, sh), sp[i], acc);
            lo = hi;
        }
    }
    acc = fme_wave_sum63(acc);
    if (tid == 63) slot[0] = acc;
}

/* copy a rectangle of nd dwords x rows from global memory (uniform base, any alignment) into LDS (rows at dst_stride, dword
 * aligned): task = 16 bytes of a row, ONE global load; the last unit of a row is moved left so that it ends with the row's last
 * dword (it overlaps its neighbour; nothing beyond the rectangle is read or written).  nd >= 4.  Two units per thread are in
 * flight before the first LDS store. */
SVT_DEV void fph_load_rect(int tid, uint8_t *dst, int dst_stride, const uint8_t *src, int src_stride, int nd, int rows) {
    const int      nu = (nd + 3) >> 2, n = nu * rows, last = 4 * nd - 16;
    const uint32_t inv = me_magics.v[nu]; /* nu >= 2 here (nd > 4); nu = 1 gives inv = 0 -> handled below */
    for (int t0 = tid; t0 < n; t0 += 2 * SVT_NT) {
        me_u32x4 v[2];
        int      o[2];
        _Pragma("unroll") for (int u = 0; u < 2; u++) {
            const int t = t0 + u * SVT_NT;
            o[u] = -1;
            if (t < n) {
                const int r = inv ? (int)__umulhi((uint32_t)t, inv) : t, i = t - ME_MUL(r, nu);
                const int b = 16 * i < last ? 16 * i : last;
                v[u] = me_ld128u_g(src + (uint32_t)(ME_MUL(r, src_stride) + b));
                o[u] = ME_MUL(r, dst_stride) + b;
            }
        }
        _Pragma("unroll") for (int u = 0; u < 2; u++)
            if (o[u] >= 0) {
                uint32_t *d = (uint32_t *)(dst + o[u]);
                d[0] = v[u].x; d[1] = v[u].y; d[2] = v[u].z; d[3] = v[u].w;
            }
    }
}

/* HME level 0: exhaustive search of sw x nr positions (sw a multiple of 16) of the 16 x 8 block (rows 0, 2, .. of the
 * 1/16-resolution SB) in the window staged at c->planes (row stride ws; window row of search row y, block row j = y + 2 j).
 * Task = 4 consecutive positions.  *key = min over (sad << 16 | (y0 + y) << 8 | x): the reference's first minimum in raster
 * order (eb_vp9_sad_loop_kernel, C_DEFAULT/EbComputeSAD_C.c:132-169).  WS != 0: the stride as a compile-time constant. */
template <int WS> SVT_DEV void fph_hme_search(const me_ctx_t *c, int tid, int ws_rt, int sw, int nr, int y0, uint32_t *key) {
    /* A task is a RUN of two groups = 8 positions (the area's width is a multiple of 16): the five operand pairs (w[i], w[i + 1]), i = 0 .. 4,
     * of a window row serve both groups -- group 0 takes pairs 0-3, group 1 pairs 1-4 -- and every pair is read ONCE (five ds_read2_b32 per
     * row for 8 positions; one group per task read its four pairs: eight per row for the same positions).  The kernel shares one LDS pipe
     * among the five workgroups of a CU and keeps it busy for 69 % of the time (tools/me_phase_lds.sh): this phase was its largest user. */
    const int      ws = WS ? WS : ws_rt, ng2 = sw >> 3, ntask = ME_MUL(ng2, nr);
    const uint32_t inv = me_magics.v[ng2]; /* ng2 in [2, 32] */
    uint32_t       b0 = 0xffffffffu, b1 = 0xffffffffu, b2 = 0xffffffffu, b3 = 0xffffffffu;
    const uint32_t *blk = (const uint32_t *)c->st->sixteenth_sb; /* the block: 8 rows of 4 dwords (same address in every lane: broadcast reads) */
    /* Which (row, run) a lane takes decides the bank conflicts of its five reads per window row: a lane reads dwords y * (ws / 4) + 2 * run + i,
     * the LDS serves 32 lanes per pass, and any two lanes of a pass on one bank with different addresses double it.  With T -> (T / 8, T % 8)
     * a pass held four CONSECUTIVE rows x eight runs -- two rows of equal parity always shared a bank: every read took 8.1 instead of 4.15 LDS
     * cycles (tools/ubench/lds_patterns.hip, profiles/r06_lds_patterns.txt).  For the common 64-wide area (eight runs per row) a pass now
     * holds rows {y, y + 1, y + 16, y + 17}: ws / 4 is odd, so row y + 16 sits 16 banks further and row y + 1 on the banks of the other
     * parity -- the four rows' 8-bank sets tile the 32 banks exactly.  Same tasks, same keys (a key carries its position). */
    const bool tile32 = ng2 == 8;
    const int  nt = tile32 ? ((nr + 31) & ~31) * 8 : ntask;
    for (int T = tid; T < nt; T += SVT_NT) {
        int y, g;
        if (tile32) {
            const int l = T & 63, blk32 = T >> 8, wv = (T >> 6) & 3;
            y = 32 * blk32 + 4 * wv + ((l >> 3) & 1) + 2 * ((l >> 5) & 1) + 16 * ((l >> 4) & 1);
            g = 2 * (l & 7);
            if (y >= nr) continue;
        } else {
            y = (int)__umulhi((uint32_t)T, inv); g = 2 * (T - ME_MUL(y, ng2));
        }
        /* byte offsets (inside the workgroup's LDS) of the run's first window dword and of the one after it, opaque to the compiler:
         * every QSAD operand pair (w[i], w[i + 1]) is then read as such by one ds_read2_b32 at an immediate offset -- the pairs of even
         * i from the first stream, of odd i from the second; nothing is assembled with register moves.  Window rows 12 and 14 lie
         * beyond the 8-bit dword offsets of ds_read2 when the stride is large: a second pair of bases serves them. */
        uint32_t o0 = (uint32_t)(c->planes - c->lds) + (uint32_t)(ME_MUL(y, ws) + 4 * g), o1 = o0 + 4, o2 = o0 + 12 * ws, o3 = o2 + 4;
        __asm__("" : "+v"(o0));
        __asm__("" : "+v"(o1));
        __asm__("" : "+v"(o2));
        __asm__("" : "+v"(o3));
        uint64_t acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0; /* group 0: even / odd rows, group 1: even / odd rows */
        _Pragma("unroll") for (int j = 0; j < 8; j++) {
            const uint8_t *wa = c->lds + (j < 6 ? o0 + 2 * j * ws : o2 + 2 * (j - 6) * ws), *wb = c->lds + (j < 6 ? o1 + 2 * j * ws : o3 + 2 * (j - 6) * ws);
            const uint64_t p0 = *(const me_u64a4 *)wa, p1 = *(const me_u64a4 *)wb, p2 = *(const me_u64a4 *)(wa + 8), p3 = *(const me_u64a4 *)(wb + 8),
                           p4 = *(const me_u64a4 *)(wa + 16);
            const uint32_t s0 = blk[4 * j], s1 = blk[4 * j + 1], s2 = blk[4 * j + 2], s3 = blk[4 * j + 3];
            uint64_t       a = (j & 1) ? acc1 : acc0, e = (j & 1) ? acc3 : acc2;
            a = svt_qsad(p0, s0, a); e = svt_qsad(p1, s0, e);
            a = svt_qsad(p1, s1, a); e = svt_qsad(p2, s1, e);
            a = svt_qsad(p2, s2, a); e = svt_qsad(p3, s2, e);
            a = svt_qsad(p3, s3, a); e = svt_qsad(p4, s3, e);
            if (j & 1) { acc1 = a; acc3 = e; } else { acc0 = a; acc2 = e; }
        }
        _Pragma("unroll") for (int u = 0; u < 2; u++) {
            const uint64_t ae = u ? acc2 : acc0, ao = u ? acc3 : acc1;
            const uint32_t lo = (uint32_t)ae + (uint32_t)ao, hi = (uint32_t)(ae >> 32) + (uint32_t)(ao >> 32); /* 8 rows x 16 x 255 < 2^16 */
            const uint32_t pos = ((uint32_t)(y0 + y) << 8) | (uint32_t)(4 * (g + u));
            /* one running minimum per position of a group; the position's offset inside the group is added at the end */
            const uint32_t k0 = (lo << 16) | pos, k1 = (lo & 0xffff0000u) | pos, k2 = (hi << 16) | pos, k3 = (hi & 0xffff0000u) | pos;
            b0 = k0 < b0 ? k0 : b0; b1 = k1 < b1 ? k1 : b1; b2 = k2 < b2 ? k2 : b2; b3 = k3 < b3 ? k3 : b3;
        }
    }
    /* (~0 stays ~0: + o cannot be allowed to wrap) */
    uint32_t k = b0;
    if (b0 != 0xffffffffu) { /* a lane either ran tasks (all four minima set) or none */
        const uint32_t m1 = b1 + 1, m2 = b2 + 2, m3 = b3 + 3;
        k = k < m1 ? k : m1; k = k < m2 ? k : m2; k = k < m3 ? k : m3;
    }
#define FME_DPP_MIN(ctrl, rmask) do { const uint32_t o_ = (uint32_t)__builtin_amdgcn_update_dpp((int)k, (int)k, (ctrl), (rmask), 0xf, false); k = o_ < k ? o_ : k; } while (0)
    FME_DPP_MIN(0x111, 0xf); FME_DPP_MIN(0x112, 0xf); FME_DPP_MIN(0x114, 0xf); FME_DPP_MIN(0x118, 0xf);
    FME_DPP_MIN(0x142, 0xa); FME_DPP_MIN(0x143, 0xc);
#undef FME_DPP_MIN
    if ((tid & 63) == 63 && k != 0xffffffffu) atomicMin(key, k);
}

/* keys -> best SAD / motion vector of every PU (curr_mv = (y << 18) | (uint16)(x << 2), :108-110) and the su_pel_enable
 * decision (:3839-4258: average vector magnitude / SAD of the 32x32 and of the 16x16 PUs against thresholds of the temporal
 * layer).  Wave 0: lanes 0-4 = PUs 0-4, lanes 16-31 = PUs 5-20 -- the four 32x32 PUs are the head of DPP row 0, the sixteen
 * 16x16 PUs are DPP row 1, so ONE row reduction yields both classes' sums; wave 1: the 64 8x8 PUs (their class is never
 * refined with cu8x8_mode 1, so its sums are not needed). */
SVT_DEV void fph_decode_gate(const me_ctx_t *c, int tid, int list, int saw, int sox, int soy) {
    me_state_t *st = c->st;
    if (tid >= 128) return;
    const int wv = ME_UNI(tid >> 6), l = tid & 63;
    const int t = wv ? 21 + l : l < 5 ? l : (l >= 16 && l < 32) ? l - 11 : -1;
    uint32_t  mv = 0, sd = 0;
    if (t >= 0) {
        const uint64_t k = st->key[t];
        const uint32_t idx = (uint32_t)k;
        if (t == 0 && list == 1) c->cand[0] = 0; /* (after the read: the table shares the keys' bytes) bi-prediction sum of the 64x64 PU */
        sd = (uint32_t)(k >> 32);
        st->best_sad[list][t] = sd;
        if (sd != (uint32_t)ME_MAX_SAD_VALUE) {
            const int yq = me_udiv((int)idx, saw);
            const int xi = (int)idx - ME_MUL(yq, saw) + sox, yi = yq + soy;
            mv = (((uint32_t)(uint16_t)yi) << 18) | (uint16_t)((uint16_t)xi << 2);
            st->best_mv[list][t] = mv;
        } else mv = st->best_mv[list][t];
    }
    if (wv == 0) {
        const int      in = (l >= 1 && l < 5) || (l >= 16 && l < 32);
        uint32_t       sx = in ? (uint32_t)(int32_t)me_mvx(mv) : 0u, sy = in ? (uint32_t)(int32_t)me_mvy(mv) : 0u, ss = in ? sd : 0u;
        sx = svt_row_prefix_add(sx, 16); sy = svt_row_prefix_add(sy, 16); ss = svt_row_prefix_add(ss, 16);
        int      x32 = __builtin_amdgcn_readlane((int)sx, 15), y32 = __builtin_amdgcn_readlane((int)sy, 15);
        uint32_t s32 = (uint32_t)__builtin_amdgcn_readlane((int)ss, 15);
        int      x16 = __builtin_amdgcn_readlane((int)sx, 31), y16 = __builtin_amdgcn_readlane((int)sy, 31);
        uint32_t s16 = (uint32_t)__builtin_amdgcn_readlane((int)ss, 31);
        uint32_t ax = (uint32_t)(x32 >> 2), ay = (uint32_t)(y32 >> 2);
        const uint32_t mag32 = ax * ax + ay * ay, sad32 = s32 >> 2;
        ax = (uint32_t)(x16 >> 4); ay = (uint32_t)(y16 >> 4);
        const uint32_t mag16 = ax * ax + ay * ay, sad16 = s16 >> 4;
        /* thresholds and tables of su_pel_enable as packed constants: thr = {48, 32, 80, 48}; t32[tl][2 a + b] bits; t16[2 a + b] */
        const int      tl = c->p->temporal_layer_index > 3 ? 3 : c->p->temporal_layer_index;
        const uint32_t thr = (0x30502030u >> (8 * tl)) & 0xffu, t2 = thr * thr;
        /* t32_: {1,0,1,0}, {1,0,1,1}, {1,0,1,0}, {1,1,1,0} -> nibbles with bit i = entry i */
        const uint32_t t32n = (0x75D5u >> (4 * tl)) & 0xfu;
        const int      i32 = 2 * !(mag32 < t2) + !(sad32 < 32 * 32 * 6), i16 = 2 * !(mag16 < t2) + !(sad16 < 16 * 16 * 2);
        const uint32_t en32 = (t32n >> i32) & 1u, en16 = (0xAu >> i16) & 1u; /* t16_ = {0, 1, 0, 1} */
        if (l == 0) *FME_GATE(st) = en32 | (en16 << 1);
    }
}

/* 16 samples at LDS byte offset a (any alignment) as four dwords */
SVT_DEV void fme_fetch16(const uint8_t *lds, uint32_t a, uint32_t v[4]) {
    const uint32_t  sh = a & 3u;
    const uint32_t *q = (const uint32_t *)(lds + (a - sh));
    const uint32_t  l0 = q[0], l1 = q[1], l2 = q[2], l3 = q[3], l4 = q[4];
    v[0] = svt_alignbyte(l1, l0, sh); v[1] = svt_alignbyte(l2, l1, sh); v[2] = svt_alignbyte(l3, l2, sh); v[3] = svt_alignbyte(l4, l3, sh);
}
/* LDS byte offset of a 16-bit source (see FME_QTAB) for a lane whose position is bp in the B / H / J planes and bf in the region
 * (both biased by -128) */
SVT_DEV uint32_t fme_src_at(uint32_t code, uint32_t bp, uint32_t bf) { return ((code & 0x8000u) ? bf : bp) + (code & 0x7fffu); }

/* ---- half- and quarter-pel refinement of the 32x32 and 16x16 PUs, both decisions and every level of the bi-prediction in ONE
 * phase: the same lane roles and the same arithmetic as ph_subpel_fast (me_core.h) -- waves 0-1 the four 32x32 PUs (32 lanes each:
 * 16 subsampled rows x 2 halves), waves 2-3 the sixteen 16x16 PUs (8 lanes each), a lane owns 16 samples of one row, DPP sums inside
 * the PU, decisions in the PU's last lane, published through LDS -- with the addressing taken out of the instruction stream:
 *   - the eight half-pel candidates lie at compile-time distances from the lane's position in plane B and fall into two alignment
 *     classes (x and x - 1): two aligned bases + shifts, every fetch an immediate offset;
 *   - the planes of a quarter-pel candidate / of a prediction come from the offset tables in LDS (FME_QTAB / FME_BTAB) instead of
 *     being derived from the packed candidate tables in every lane (plane, stride, displacement: ~15 instructions per source);
 *   - the 64x64 PU's vector is wave-uniform: its planes and strides are chosen on the scalar unit;
 *   - candidate sums stay packed two per dword up to the last cross-row step of a 32x32 PU. */
SVT_DEV void fph_subpel(const me_ctx_t *c, int tid, int list, int sox, int soy, int en32, int en16, int bipred, uint32_t *pr) {
    me_state_t *st = c->st;
    const int   w = ME_UNI(tid >> 6), l = tid & 63, big = w < 2;
    const int   refine = big ? en32 : en16; /* wave-uniform */
    if (!refine && !bipred) return;
    const int rs = c->L.region_stride, ps = c->L.plane_stride, pb = c->L.plane_bytes;
    int       pu, n, r, xo, px, py, last;
    if (big) { pu = 1 + 2 * w + (l >> 5); n = pu; r = (l & 31) >> 1; xo = (l & 1) * 16; px = ((pu - 1) & 1) * 32; py = ((pu - 1) >> 1) * 32; last = (l & 31) == 31; }
    else { pu = 5 + 8 * (w - 2) + (l >> 3); n = me_z4(pu - 5) + 5; r = l & 7; xo = 0; px = ((pu - 5) & 3) * 16; py = ((pu - 5) >> 2) * 16; last = (l & 7) == 7; }
    uint32_t s[4];
    {
        const uint32_t *sp = (const uint32_t *)(c->src + (py + 2 * r) * ME_SB + px + xo);
        s[0] = sp[0]; s[1] = sp[1]; s[2] = sp[2]; s[3] = sp[3];
    }
    /* the lane's sample position relative to the search area's origin, and the LDS byte offsets of sample (0, 0) of the region /
     * of plane B (biased by -128 for the table offsets) */
    const int      lx = px + xo - sox, ly = py + 2 * r - soy;
    const uint32_t of = (uint32_t)(c->region - c->lds) + (uint32_t)(ME_RGN_GY * rs + ME_RGN_GX) - 128u,
                   op = (uint32_t)(c->planes - c->lds) + (uint32_t)(ME_PL_G * ps + ME_PL_G) - 128u;
    uint32_t mv = st->best_mv[list][n], best = st->best_sad[list][n];
    int      xm = me_mvx(mv), ym = me_mvy(mv);
    /* ---- half-pel: 8 candidates (pu_half_pel_refinement :1076-1559) ---- */
    if (refine) {
        const int xs = (int16_t)(xm >> 2) + lx, ys = (int16_t)(ym >> 2) + ly;
        const uint32_t a0 = op + 128u + (uint32_t)(ME_MUL(ys, ps) + xs); /* plane B at (xs, ys) */
        uint32_t       sh0 = a0 & 3u, q0 = a0 - sh0, sh1 = (a0 - 1u) & 3u, q1 = a0 - 1u - sh1;
        __asm__("" : "+v"(q0));
        __asm__("" : "+v"(q1));
        uint32_t d[8];
        /* The candidates one sample apart in the same plane row -- L / R (plane B), TL / TR and BL / BR (plane J) -- lie in the SAME five
         * dwords from the aligned address below the left one: a byte selector per side (v_perm_b32 over a dword pair, offsets 0 .. 4)
         * takes both from one fetch.  15 LDS reads for the eight candidates instead of 24 (the kernel's LDS pipe is as busy as its vector
         * unit: tools/me_phase_lds.sh); T / B (plane H, two rows) keep their own fetches. */
        const uint32_t selL = 0x03020100u + 0x01010101u * sh1, selR = selL + 0x01010101u;
        _Pragma("unroll") for (int i = 0; i < 8; i++) {
            int hpl, hdx, hdy;
            me_hcand_get(i, &hpl, &hdx, &hdy);
            if (hpl == 2) { /* T, B */
                const uint32_t *q = (const uint32_t *)(c->lds + q0 + (uint32_t)((hpl - 1) * pb + hdy * ps));
                const uint32_t  l0 = q[0], l1 = q[1], l2 = q[2], l3 = q[3], l4 = q[4];
                uint32_t        t = svt_sad4(svt_alignbyte(l1, l0, sh0), s[0], 0);
                t = svt_sad4(svt_alignbyte(l2, l1, sh0), s[1], t);
                t = svt_sad4(svt_alignbyte(l3, l2, sh0), s[2], t);
                d[i] = svt_sad4(svt_alignbyte(l4, l3, sh0), s[3], t);
            } else if (hdx) { /* the left one of a pair: fetches for both; its right neighbour is the candidate with the same plane and row */
                int ir = -1;
                _Pragma("unroll") for (int k = 0; k < 8; k++) {
                    int kp, kx, ky;
                    me_hcand_get(k, &kp, &kx, &ky);
                    if (kp == hpl && ky == hdy && !kx) ir = k;
                }
                const uint32_t *q = (const uint32_t *)(c->lds + q1 + (uint32_t)((hpl - 1) * pb + hdy * ps));
                const uint32_t  l0 = q[0], l1 = q[1], l2 = q[2], l3 = q[3], l4 = q[4];
                uint32_t        t = svt_sad4((uint32_t)__builtin_amdgcn_perm(l1, l0, selL), s[0], 0), u = svt_sad4((uint32_t)__builtin_amdgcn_perm(l1, l0, selR), s[0], 0);
                t = svt_sad4((uint32_t)__builtin_amdgcn_perm(l2, l1, selL), s[1], t); u = svt_sad4((uint32_t)__builtin_amdgcn_perm(l2, l1, selR), s[1], u);
                t = svt_sad4((uint32_t)__builtin_amdgcn_perm(l3, l2, selL), s[2], t); u = svt_sad4((uint32_t)__builtin_amdgcn_perm(l3, l2, selR), s[2], u);
                d[i] = svt_sad4((uint32_t)__builtin_amdgcn_perm(l4, l3, selL), s[3], t);
                d[ir] = svt_sad4((uint32_t)__builtin_amdgcn_perm(l4, l3, selR), s[3], u);
            }
        }
        /* a lane's sums stay below 2^12, 16 lanes' below 2^16: two candidates per dword up to the row; a 32x32 PU's second row is
         * added half by half into 32-bit sums */
        _Pragma("unroll") for (int i = 0; i < 4; i++) {
            uint32_t v = d[i] | (d[i + 4] << 16);
            v = SVT_DPP_ADD(v, 0x111); v = SVT_DPP_ADD(v, 0x112); v = SVT_DPP_ADD(v, 0x114);
            if (big) {
                v = SVT_DPP_ADD(v, 0x118);
                const uint32_t x = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false); /* row_bcast:15 into rows 1 and 3 */
                d[i] = (v & 0xffffu) + (x & 0xffffu); d[i + 4] = (v >> 16) + (x >> 16);
            } else { d[i] = v & 0xffffu; d[i + 4] = v >> 16; }
        }
        /* decisions (meaningful in the PU's last lane): strict '<' in test order = minimum of (2 d) << 3 | test index; direction by
         * the tie order L,R,T,B,TL,TR,BL,BR (:1531-1556) */
        uint32_t k[8];
        _Pragma("unroll") for (int i = 0; i < 8; i++) k[i] = (d[i] << 4) | (uint32_t)i;
        const uint32_t m5 = me_min3(me_min3(k[0], k[1], k[2]), k[3], k[4]);
        const uint32_t km = me_min3(m5 < k[5] ? m5 : k[5], k[6], k[7]);
        const uint32_t kr = me_min3(m5 < k[5] ? m5 : k[5], k[6] + 1, k[7] - 1); /* ranks of candidates 6 (BR) and 7 (BL) swapped */
        if ((km >> 3) < best) {
            int sx, sy;
            me_dmv_get((int)(km & 7u), &sx, &sy);
            best = km >> 3; mv = me_pack_mv(xm + 2 * sx, ym + 2 * sy);
        }
        const uint32_t dir = (0x46205137u >> (4 * (kr & 7u))) & 7u; /* tie rank L,R,T,B,TL,TR,BL,BR -> direction code */
        if (last) { st->best_sad[list][n] = best; st->best_mv[list][n] = mv; st->dir[n] = (uint8_t)dir; }
    }
    __asm__ volatile("" ::: "memory"); /* the reads below must stay behind the stores above (other lanes' data) */
    /* ---- quarter-pel: the three positions around the half-pel direction (pu_quarter_pel_refinement_on_the_fly :2471-2715) ---- */
    if (refine) {
        mv = st->best_mv[list][n]; best = st->best_sad[list][n];
        const int dir = st->dir[n];
        xm = me_mvx(mv); ym = me_mvy(mv);
        const int      method = (ym & 2) + ((xm & 2) >> 1);
        const int      xs = (int16_t)((xm + 2) >> 2) + lx, ys = (int16_t)((ym + 2) >> 2) + ly;
        const uint32_t bp = op + (uint32_t)(ME_MUL(ys, ps) + xs), bf = of + (uint32_t)(ME_MUL(ys, rs) + xs);
        const int      d0 = method != 0 ? dir ^ 4 : dir;
        uint32_t       q[3], pos[3], e[3];
        _Pragma("unroll") for (int j = 0; j < 3; j++) {
            pos[j] = (0x07361524u >> (4 * ((d0 + j - 1) & 7))) & 7u; /* direction code -> L,R,T,B,TL,TR,BR,BL index */
            e[j] = FME_QTAB(st)[method * 8 + (int)pos[j]];
        }
        _Pragma("unroll") for (int j = 0; j < 3; j++) {
            uint32_t va[4], vb[4];
            fme_fetch16(c->lds, fme_src_at(e[j] & 0xffffu, bp, bf), va);
            fme_fetch16(c->lds, fme_src_at(e[j] >> 16, bp, bf), vb);
            uint32_t t = svt_sad4(svt_avg4(va[0], vb[0]), s[0], 0);
            t = svt_sad4(svt_avg4(va[1], vb[1]), s[1], t);
            t = svt_sad4(svt_avg4(va[2], vb[2]), s[2], t);
            q[j] = svt_sad4(svt_avg4(va[3], vb[3]), s[3], t);
        }
        uint32_t v01 = q[0] | (q[1] << 16), v2 = q[2];
        v01 = SVT_DPP_ADD(v01, 0x111); v01 = SVT_DPP_ADD(v01, 0x112); v01 = SVT_DPP_ADD(v01, 0x114);
        v2 = SVT_DPP_ADD(v2, 0x111); v2 = SVT_DPP_ADD(v2, 0x112); v2 = SVT_DPP_ADD(v2, 0x114);
        if (big) {
            v01 = SVT_DPP_ADD(v01, 0x118); v2 = SVT_DPP_ADD(v2, 0x118);
            const uint32_t x = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v01, 0x142, 0xa, 0xf, false);
            q[0] = (v01 & 0xffffu) + (x & 0xffffu); q[1] = (v01 >> 16) + (x >> 16);
            q[2] = v2 + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v2, 0x142, 0xa, 0xf, false);
        } else { q[0] = v01 & 0xffffu; q[1] = v01 >> 16; q[2] = v2; }
        const uint32_t km = me_min3((q[0] << 4) | pos[0], (q[1] << 4) | pos[1], (q[2] << 4) | pos[2]);
        if (last && (km >> 3) < best) {
            int sx, sy;
            me_dmv_get((int)(km & 7u), &sx, &sy);
            st->best_sad[list][n] = km >> 3; st->best_mv[list][n] = me_pack_mv(xm + sx, ym + sy);
        }
    }
    /* ---- the lane's 16 samples of the PU's prediction at its final motion vector: kept in registers after list 0; after list 1
     * averaged with them and compared with the source -- the PU's bi-prediction distortion (bi_pred_averging :3466-3560), summed
     * over the PU's lanes, written by its last lane ---- */
    if (bipred) {
        __asm__ volatile("" ::: "memory");
        {
            const uint32_t fm = st->best_mv[list][n];
            const int      mx = me_mvx(fm), my = me_mvy(fm);
            const int      xi = (int16_t)(mx >> 2) + lx, yi = (int16_t)(my >> 2) + ly;
            const uint32_t e = FME_BTAB(st)[(mx & 3) + ((my & 3) << 2)];
            const uint32_t bp = op + (uint32_t)(ME_MUL(yi, ps) + xi), bf = of + (uint32_t)(ME_MUL(yi, rs) + xi);
            uint32_t       va[4], vb[4];
            fme_fetch16(c->lds, fme_src_at(e & 0xffffu, bp, bf), va);
            fme_fetch16(c->lds, fme_src_at(e >> 16, bp, bf), vb);
            _Pragma("unroll") for (int i = 0; i < 4; i++) va[i] = svt_avg4(va[i], vb[i]); /* one plane: avg(a, a) = a */
            if (list == 0) { pr[4] = va[0]; pr[5] = va[1]; pr[6] = va[2]; pr[7] = va[3]; }
            else {
                uint32_t d = svt_sad4(svt_avg4(pr[4], va[0]), s[0], 0);
                d = svt_sad4(svt_avg4(pr[5], va[1]), s[1], d);
                d = svt_sad4(svt_avg4(pr[6], va[2]), s[2], d);
                d = svt_sad4(svt_avg4(pr[7], va[3]), s[3], d);
                d = me_pu_lanes_sum(d, big);
                if (last) c->cand[pu] = d;
            }
        }
        /* the 64x64 PU (never refined on this path: its vector is the full-pel one, the same in every lane -> planes, strides and
         * alignment are scalars): every lane takes dword tid & 15 of the subsampled rows 2 (tid >> 4) and 2 (tid >> 4) + 32; wave
         * sums into cand[0], which fph_decode_gate zeroed */
        {
            const uint32_t mv0 = (uint32_t)ME_UNI(st->best_mv[list][0]);
            const int      mx = me_mvx(mv0), my = me_mvy(mv0);
            const int      xi = (int16_t)(mx >> 2) - sox, yi = (int16_t)(my >> 2) - soy;
            int            has_b;
            const uint32_t e = me_btab_get((mx & 3) + ((my & 3) << 2), &has_b);
            const int      pa = (int)(e & 3), pbn = (int)((e >> 4) & 3);
            const uint8_t *a = me_plane_at(c, pa, xi + (int)((e >> 2) & 1), yi + (int)((e >> 3) & 1));
            const uint8_t *b = me_plane_at(c, pbn, xi + (int)((e >> 6) & 1), yi + (int)((e >> 7) & 1));
            const int      sa = me_plane_stride(c, pa), sb = me_plane_stride(c, pbn);
            uint32_t       d0 = 0;
            _Pragma("unroll") for (int k = 0; k < 2; k++) {
                const int rr = 2 * (tid >> 4) + 32 * k, ii = tid & 15;
                uint32_t  vb = me_ld32u(a + ME_MUL(rr, sa) + 4 * ii);
                if (has_b) vb = svt_avg4(vb, me_ld32u(b + ME_MUL(rr, sb) + 4 * ii));
                if (list == 0) pr[k] = vb;
                else d0 = svt_sad4(svt_avg4(pr[k], vb), *(const uint32_t *)(c->src + rr * ME_SB + 4 * ii), d0);
            }
            if (list != 0) {
                d0 = fme_wave_sum63(d0);
                if (l == 63) atomicAdd(&c->cand[0], d0);
            }
        }
    }
}

template <int SPEC> __device__ __forceinline__ void me_sb_run_fast(const me_ctx_t *c, int tid_) {
    int tid = tid_;
    const svt_me_params *p = c->p;
    me_state_t          *st = c->st;
    const int            nlist = p->num_ref_lists;
    uint32_t             pred0_regs[16]; /* list 0 prediction dwords of this lane's bi-prediction work (ph_subpel_fast) */
    const int            do_hme = c->sb_h == ME_SB;
    const int            ox = (int16_t)c->sb_x, oy = (int16_t)c->sb_y;

    ME_PHASE(fph_init(c, tid, do_hme));
    FME_MARK(0);

    for (int list = 0; list < nlist; list++) {
        const svt_plane *rf = &c->pic->ref[list].full; /* descriptor fields come by scalar loads where they are used */
        int16_t          xsc = 0, ysc = 0;
        uint32_t         zero_c = 0; /* 2 * SAD of the block at (0, 0) of this list, when test_search_area_bounds ran */
        int              have_zero = 0;
        if (p->temporal_layer_index > 0 || list == 0) {
            /* ---- test_search_area_bounds ---- */
            {
                const int pad = ME_SB - 1, W = rf->width, H = rf->height, rstride = rf->stride;
                const int tw = p->hme_level0_total_search_area_width, th = p->hme_level0_total_search_area_height;
                int16_t   dirx = 0, diry = 0;
                int32_t   off[5];
                off[0] = 0;
                off[1] = ME_MUL(me_clip_center(oy, 0, pad, H), rstride) + me_clip_center(ox, (int16_t)tw, pad, W);
                off[2] = ME_MUL(me_clip_center(oy, (int16_t)(0 - th), pad, H), rstride) + me_clip_center(ox, 0, pad, W);
                off[3] = ME_MUL(me_clip_center(oy, (int16_t)th, pad, H), rstride) + me_clip_center(ox, 0, pad, W);
                off[4] = 0;
                int nc = 4;
                if (list == 1) {
                    const uint32_t mv00 = (uint32_t)ME_UNI(st->best_mv[0][0]);
                    dirx = (int16_t)(0 - (me_mvx(mv00) >> 2));
                    diry = (int16_t)(0 - (me_mvy(mv00) >> 2));
                    off[4] = ME_MUL(me_clip_center(oy, diry, pad, H), rstride) + me_clip_center(ox, dirx, pad, W);
                    nc = 5;
                }
                ME_PHASE(fph_center(c, tid, me_pix(rf, c->sb_x, c->sb_y), rstride, nc, off, FME_ACC(st, 2 * list)));
                uint32_t s[5];
                fme_center_sums(FME_ACC(st, 2 * list), nc, s);
                zero_c = s[0] << 1; have_zero = 1;
                const uint32_t b_c = s[1] << 1, c_c = s[2] << 1, d_c = s[3] << 1;
                const uint32_t a_c = zero_c; /* [quirk] A is evaluated at the zero-MV address (:4302-4327) */
                const uint32_t dir_c = list == 1 ? s[4] << 1 : 0xffffffffu;
                uint32_t       best = zero_c;
                if (a_c < best) best = a_c;
                if (b_c < best) best = b_c;
                if (c_c < best) best = c_c;
                if (d_c < best) best = d_c;
                if (dir_c < best) best = dir_c;
                if (best == zero_c) { xsc = 0; ysc = 0; }
                else if (best == a_c) { xsc = (int16_t)(0 - tw); ysc = 0; }
                else if (best == b_c) { xsc = (int16_t)tw; ysc = 0; }
                else if (best == c_c) { xsc = 0; ysc = (int16_t)(0 - th); }
                else if (best == dir_c) { xsc = list ? dirx : 0; ysc = list ? diry : 0; }
                else { xsc = 0; ysc = (int16_t)th; }
            }
            FME_MARK(1);
            /* ---- HME level 0, one region (hme_level0 / the single-quadrant form, :2717-2760, :2872-2920) ---- */
            if (do_hme) {
                const svt_plane *r16 = &c->pic->ref[list].sixteenth;
                const int        gox = (int16_t)(c->sb_x >> 2), goy = (int16_t)(c->sb_y >> 2);
                int16_t          w = c->L.hme_tw0, h = c->L.hme_th0;
                int16_t          hx = (int16_t)(-(int16_t)(w >> 1) + (int16_t)(xsc >> 2)), hy = (int16_t)(-(int16_t)(h >> 1) + (int16_t)(ysc >> 2));
                me_clip_area(gox, &hx, &w, r16->origin_x - 1, r16->width);
                me_clip_area(goy, &hy, &h, r16->origin_y - 1, r16->height);
                if ((w & 15) != 0) w = (int16_t)((w >> 4) << 4);
                int bx = (int16_t)(xsc >> 2), by = (int16_t)(ysc >> 2); /* the position that stands when nothing is searched */
                if (w > 0 && h > 0) {
                    const int wbytes = w + 16 + 3, nd = (wbytes + 3) >> 2;
                    int       ws = ((wbytes + 3) & ~3) + 4;
                    if (((ws >> 2) & 1) == 0) ws += 4;
                    const int band = c->pic->hme_band; /* search rows whose window fits the scratch (host: widest window of the picture) */
                    for (int y0 = 0; y0 < h; y0 += band) {
                        const int nr = h - y0 < band ? h - y0 : band;
                        ME_PHASE(fph_load_rect(tid, c->planes, ws, me_pix(r16, gox + hx, goy + hy + y0), r16->stride, nd, nr + 14));
                        if (ws == 92) ME_PHASE(fph_hme_search<92>(c, tid, ws, w, nr, y0, FME_HKEY(st, list)));
                        else ME_PHASE(fph_hme_search<0>(c, tid, ws, w, nr, y0, FME_HKEY(st, list)));
                    }
                    const uint32_t k = (uint32_t)ME_UNI(*FME_HKEY(st, list));
                    if (k != 0xffffffffu) { bx = (int)(k & 0xffu); by = (int)((k >> 8) & 0xffu); }
                }
                xsc = (int16_t)((int16_t)(bx + hx) * 4); ysc = (int16_t)((int16_t)(by + hy) * 4);
            }
            FME_MARK(2);
        }

        int16_t saw, sah, sox, soy;
        int     W, H, w8, tail_extra, loaded = 0;
        /* search area of a centre: position, clipping, derived sizes (Codec/EbMotionEstimation.c:5008-5060) */
#define FME_SET_AREA()                                                                                              \
    do {                                                                                                            \
        saw = (int16_t)(p->search_area_width < 127 ? p->search_area_width : 127);                                    \
        sah = (int16_t)(p->search_area_height < 127 ? p->search_area_height : 127);                                  \
        sox = (int16_t)(xsc - (saw >> 1)); soy = (int16_t)(ysc - (sah >> 1));                                        \
        me_clip_area(ox, &sox, &saw, ME_SB - 1, c->pic_w);                                                           \
        me_clip_area(oy, &soy, &sah, ME_SB - 1, c->pic_h);                                                           \
        W = saw + ME_SB - 1; H = sah + ME_SB - 1; w8 = saw - (saw & 7); tail_extra = (saw & 7) ? 16 : 0;             \
    } while (0)
        /* stage the search region (+ halo) of this list in LDS; the full-pel keys are reset on the way */
#define FME_LOAD_REGION()                                                                                           \
    ME_PHASE(fph_load_rect(tid, c->region, c->L.region_stride, me_pix(rf, c->sb_x + sox - ME_RGN_GX, c->sb_y + soy - ME_RGN_GY), \
                           rf->stride, (W + ME_RGN_GX + 4 + tail_extra + 3) >> 2, H + 2 * ME_RGN_GY + 1);            \
             if (tid < 85) st->key[tid] = ((uint64_t)ME_MAX_SAD_VALUE << 32))
        if (xsc != 0 || ysc != 0) {
            /* ---- check_zero_zero_center (:4420-4500): the centre found above against (0, 0).  The SAD at (0, 0) is the one
             * test_search_area_bounds produced for this list; the SAD at the centre is taken from the search region, which is
             * staged around that centre first ---- */
            xsc = me_clip_center(ox, xsc, ME_SB - 1, rf->width);
            ysc = me_clip_center(oy, ysc, ME_SB - 1, rf->height);
            FME_SET_AREA();
            const int col = ME_RGN_GX + (xsc - sox), row = ME_RGN_GY + (ysc - soy);
            const int inside = have_zero && col >= 0 && row >= 0 && col + ME_SB <= W + ME_RGN_GX + 4 + tail_extra && row + c->sb_h <= H + 2 * ME_RGN_GY + 1;
            uint32_t  z, hh;
            if (inside) {
                FME_LOAD_REGION();
                ME_PHASE(fph_region_center(c, tid, col, row, FME_ACC(st, 2 * list + 1)));
                z = zero_c; hh = (uint32_t)ME_UNI(*FME_ACC(st, 2 * list + 1)) << 1;
                loaded = 1;
            } else {
                int32_t off[5] = {0, ME_MUL(ysc, rf->stride) + xsc, 0, 0, 0};
                ME_PHASE(fph_center(c, tid, me_pix(rf, c->sb_x, c->sb_y), rf->stride, 2, off, FME_ACC(st, 2 * list + 1)));
                uint32_t s[5];
                fme_center_sums(FME_ACC(st, 2 * list + 1), 2, s);
                z = s[0] << 1; hh = s[1] << 1;
            }
            if (z <= hh) { xsc = 0; ysc = 0; loaded = 0; } /* min(z, h) == z */
        }
        FME_MARK(3);
        if (!loaded) {
            FME_SET_AREA();
            FME_LOAD_REGION();
        }
#undef FME_SET_AREA
#undef FME_LOAD_REGION
        FME_MARK(4);
        /* ---- full-pel search ---- */
        {
            uint32_t *U = (uint32_t *)c->planes;
            if (saw >= 8 && (saw & 7) == 0 && saw * sah <= 4096) {
                ME_PHASE(ph_fullpel_fused(c, tid, saw, sah, 0));
                FME_MARK(5);
                FME_MARK(6);
            } else { /* clipped at a picture border: the table form, in chunks of search rows (me_sb_run) */
                const int max_pos = c->L.scratch_bytes / (4 * ME_PU_STRIDE);
                int       rows_chunk = max_pos / saw;
                if (rows_chunk < 1) rows_chunk = 1;
                if (rows_chunk > sah) rows_chunk = sah;
                for (int y0 = 0; y0 < sah; y0 += rows_chunk) {
                    const int ny = y0 + rows_chunk <= sah ? rows_chunk : sah - y0;
                    ME_PHASE(ph_fullpel_sad8(c, tid, U, saw, y0, ny, w8));
                    ME_PHASE(ph_fullpel_sum16(c, tid, U, saw, ny, w8));
                    ME_PHASE(ph_fullpel_sum32(c, tid, U, ny * saw));
                    ME_PHASE(ph_fullpel_argmin(c, tid, U, saw, y0, ny));
                }
            }
            /* (the interpolation of the half-pel planes reads the region alone: it shares the phase -- and the barrier -- of the key decode) */
            ME_PHASE(fph_decode_gate(c, tid, list, saw, sox, soy); ph_interp_strips(c, tid, W, H));
        }
        FME_MARK(7);
        const uint32_t gate = (uint32_t)ME_UNI(*FME_GATE(st));
        const int      en32 = (int)(gate & 1u), en16 = (int)((gate >> 1) & 1u);
        FME_MARK(8);
        FME_MARK(9);
        /* SUB_SAD refinement of the 32x32 / 16x16 PUs, both decisions and every level of the bi-prediction: one phase */
        const int fast_bi = nlist == 2;
        if (en32 || en16 || fast_bi) ME_PHASE(fph_subpel(c, tid, list, sox, soy, en32, en16, fast_bi, pred0_regs));
        FME_MARK(10);
        FME_MARK(11);
        FME_MARK(12);
    }

    /* ---- results ---- */
    uint32_t *ow = (uint32_t *)c->planes;
    ME_PHASE(ph_output(c, tid, 0, ow));
    {
        uint32_t SVT_GLOBAL *g = SVT_AS_GLOBAL(uint32_t, c->pic->results + (size_t)c->sb_index * 85);
        for (int t = tid; t < 850; t += SVT_NT) g[t] = ow[t];
        if (c->pic->rcme && tid == 0) {
            uint32_t acc = 0;
            for (int i = 0; i < 16; i++) acc += ow[(5 + i) * 10 + 2];
            *SVT_AS_GLOBAL(uint32_t, &c->pic->rcme[c->sb_index]) = acc;
        }
    }
    FME_MARK(13);
}

#endif /* !SVT_HOST_EMU */
#endif /* SVT_ME_FAST_H */
