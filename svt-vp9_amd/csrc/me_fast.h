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

/* accumulators of the centre tests: slot s = 2 * list + (0: test_search_area_bounds, 1: check_zero_zero_center), two 64-bit
 * words each, zeroed once per SB; they live in the bytes of hme_sad (the general driver's per-level results: unused here) */
#ifdef ME_ASM_MARKS /* static instruction counts (tools/me_static_counts.py): a comment in the assembly at every mark */
#define FME_MARK(i) __asm__ volatile("; @MARK %0" ::"n"(i))
#else
#define FME_MARK(i) ((void)0)
#endif
#define FME_ACC(st, s) ((unsigned long long *)(st)->hme_sad + 2 * (s))
/* HME arg-min key (sad << 16 | y << 8 | x) of list l: in the bytes of hme_keys */
#define FME_HKEY(st, l) ((uint32_t *)(st)->hme_keys + (l))
/* su_pel_enable decision of the current list: bit 0 = 32x32, bit 1 = 16x16 */
#define FME_GATE(st) (&(st)->supel[0])

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
    if (tid >= 128 && tid < 144) ((uint32_t *)st->hme_sad)[tid - 128] = 0;
    if (tid >= 160 && tid < 162) FME_HKEY(st, tid - 160)[0] = 0xffffffffu;
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
 * (test_search_area_bounds :4260 / check_zero_zero_center :3758): thread = one (row, 8 bytes) piece of every candidate;
 * base + off[k] = top-left sample of candidate k.  The sums of the workgroup are ADDED to slot[0] (candidates 0-2 at bits 0 /
 * 20 / 40) and slot[1] (candidates 3, 4 at bits 0 / 20). */
SVT_DEV void fph_center(const me_ctx_t *c, int tid, const uint8_t *base, int rstride, int nc, const int32_t *off, unsigned long long *slot) {
    const int r = tid >> 3, i = tid & 7;
    uint32_t  acc[5] = {0, 0, 0, 0, 0};
    if (2 * r < c->sb_h) {
        const uint32_t voff = (uint32_t)(ME_MUL(2 * r, rstride) + 8 * i);
        me_u32x2       v[5];
        _Pragma("unroll") for (int k = 0; k < 5; k++)
            if (k < nc) v[k] = me_ld64u_g(base + off[k] + voff);
        const uint32_t *sp = (const uint32_t *)(c->src + (2 * r) * ME_SB + 8 * i);
        const uint32_t  s0 = sp[0], s1 = sp[1];
        _Pragma("unroll") for (int k = 0; k < 5; k++)
            if (k < nc) acc[k] = svt_sad4(v[k].y, s1, svt_sad4(v[k].x, s0, 0));
    }
    const uint32_t p0 = fme_half_sums_pk(acc[0] | (acc[1] << 16)), p1 = fme_half_sums_pk(acc[2] | (acc[3] << 16)),
                   p2 = nc > 4 ? fme_half_sums_pk(acc[4]) : 0u;
    const uint32_t a0 = (uint32_t)__builtin_amdgcn_readlane((int)p0, 31), b0 = (uint32_t)__builtin_amdgcn_readlane((int)p0, 63);
    const uint32_t a1 = (uint32_t)__builtin_amdgcn_readlane((int)p1, 31), b1 = (uint32_t)__builtin_amdgcn_readlane((int)p1, 63);
    const uint32_t a2 = nc > 4 ? (uint32_t)__builtin_amdgcn_readlane((int)p2, 31) : 0u, b2 = nc > 4 ? (uint32_t)__builtin_amdgcn_readlane((int)p2, 63) : 0u;
    const unsigned long long x = (unsigned long long)((a0 & 0xffffu) + (b0 & 0xffffu)) | ((unsigned long long)((a0 >> 16) + (b0 >> 16)) << 20) |
                                 ((unsigned long long)((a1 & 0xffffu) + (b1 & 0xffffu)) << 40);
    const unsigned long long y = (unsigned long long)((a1 >> 16) + (b1 >> 16)) | ((unsigned long long)(a2 + b2) << 20);
    if ((tid & 63) == 0) { atomicAdd(&slot[0], x); atomicAdd(&slot[1], y); }
}
/* the five sums back as scalars (after the barrier that follows fph_center) */
SVT_DEV void fme_center_sums(const unsigned long long *slot, uint32_t s[5]) {
    const unsigned long long x = slot[0], y = slot[1];
    const uint32_t xl = (uint32_t)ME_UNI((uint32_t)x), xh = (uint32_t)ME_UNI((uint32_t)(x >> 32)), yl = (uint32_t)ME_UNI((uint32_t)y),
                   yh = (uint32_t)ME_UNI((uint32_t)(y >> 32));
    s[0] = xl & 0xfffffu; s[1] = ((xl >> 20) | (xh << 12)) & 0xfffffu; s[2] = (xh >> 8) & 0xfffffu;
    s[3] = yl & 0xfffffu; s[4] = ((yl >> 20) | (yh << 12)) & 0xfffffu;
}

/* the same SAD for ONE displaced block that already sits in the LDS search region (region byte (col, row) = its top-left
 * sample): added to the low word of slot[0] */
SVT_DEV void fph_region_center(const me_ctx_t *c, int tid, int col, int row, unsigned long long *slot) {
    const int r = tid >> 3, i = tid & 7, rs = c->L.region_stride;
    uint32_t  acc = 0;
    if (2 * r < c->sb_h) {
        const uint8_t  *p = c->region + ME_MUL(row + 2 * r, rs) + col + 8 * i;
        const uint32_t  sh = (uint32_t)((uintptr_t)p & 3);
        const uint32_t *q = (const uint32_t *)(p - sh);
        const uint32_t  l0 = q[0], l1 = q[1], l2 = q[2];
        const uint32_t *sp = (const uint32_t *)(c->src + (2 * r) * ME_SB + 8 * i);
        acc = svt_sad4(svt_alignbyte(l2, l1, sh), sp[1], svt_sad4(svt_alignbyte(l1, l0, sh), sp[0], 0));
    }
    acc = fme_wave_sum63(acc);
    if ((tid & 63) == 63) atomicAdd((uint32_t *)slot, acc);
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
    const int      ws = WS ? WS : ws_rt, ng = sw >> 2, ntask = ME_MUL(ng, nr);
    const uint32_t inv = me_magics.v[ng]; /* ng in [4, 64] */
    uint32_t       b0 = 0xffffffffu, b1 = 0xffffffffu, b2 = 0xffffffffu, b3 = 0xffffffffu;
    for (int T = tid; T < ntask; T += SVT_NT) {
        const int y = (int)__umulhi((uint32_t)T, inv), g = T - ME_MUL(y, ng);
        uint32_t  lo, hi;
        me_qsad_16x8(c->st->sixteenth_sb, c->planes + ME_MUL(y, ws) + 4 * g, ws, &lo, &hi);
        const uint32_t pos = ((uint32_t)(y0 + y) << 8) | (uint32_t)(4 * g);
        /* one running minimum per position of the group; the position's offset inside the group is added at the end */
        const uint32_t k0 = (lo << 16) | pos, k1 = (lo & 0xffff0000u) | pos, k2 = (hi << 16) | pos, k3 = (hi & 0xffff0000u) | pos;
        b0 = k0 < b0 ? k0 : b0; b1 = k1 < b1 ? k1 : b1; b2 = k2 < b2 ? k2 : b2; b3 = k3 < b3 ? k3 : b3;
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
                fme_center_sums(FME_ACC(st, 2 * list), s);
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
                z = zero_c; hh = (uint32_t)ME_UNI(*(const uint32_t *)FME_ACC(st, 2 * list + 1)) << 1;
                loaded = 1;
            } else {
                int32_t off[5] = {0, ME_MUL(ysc, rf->stride) + xsc, 0, 0, 0};
                ME_PHASE(fph_center(c, tid, me_pix(rf, c->sb_x, c->sb_y), rf->stride, 2, off, FME_ACC(st, 2 * list + 1)));
                uint32_t s[5];
                fme_center_sums(FME_ACC(st, 2 * list + 1), s);
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
                ME_PHASE(ph_fullpel_fused(c, tid, saw, sah));
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
            ME_PHASE(fph_decode_gate(c, tid, list, saw, sox, soy));
        }
        FME_MARK(7);
        const uint32_t gate = (uint32_t)ME_UNI(*FME_GATE(st));
        const int      en32 = (int)(gate & 1u), en16 = (int)((gate >> 1) & 1u);
        FME_MARK(8);
        ME_PHASE(ph_interp_strips(c, tid, W, H));
        FME_MARK(9);
        /* SUB_SAD refinement of the 32x32 / 16x16 PUs, both decisions and every level of the bi-prediction: one phase */
        const int fast_bi = nlist == 2;
        if (en32 || en16 || fast_bi) ME_PHASE(ph_subpel_fast(c, tid, list, sox, soy, en32, en16, fast_bi, pred0_regs));
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
