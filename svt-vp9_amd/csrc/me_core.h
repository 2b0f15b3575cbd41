/*
 * me_core.h -- motion estimation of one 64x64 super-block by one 256-thread workgroup (CDNA4 / gfx950).
 *
 * Implements, bit-exactly, the per-SB flow of the reference's motion_estimate_sb
 * (Source/Lib/Codec/EbMotionEstimation.c:4524-5305) for the C_DEFAULT kernels:
 *   test_search_area_bounds (:4260) -> HME level 0/1/2 (:2717-3308, eb_vp9_sad_loop_kernel
 *   C_DEFAULT/EbComputeSAD_C.c:132) -> check_zero_zero_center (:3758) -> full_pel_search_sb (:951,
 *   C_DEFAULT/EbComputeSAD_C.c:193-375, EbMeSadCalculation_C.c:16-99) -> su_pel_enable (:3839) ->
 *   interpolate_search_region_avc (:992, C_DEFAULT/EbAvcStyleMcp_C.c) -> half_pel_search_sb (:1565) ->
 *   quarter_pel_search_sb (:2471) -> bi_prediction_search (:3695) -> candidate ordering (:5186-5293).
 *
 * MI355X mapping
 *   - one workgroup (4 wave64) per SB; both reference lists are processed inside the workgroup because
 *     list 1 starts from list 0's 64x64 motion vector (:4450);
 *   - the 64x64 source SB, the reference search region (+ halo) and its three half-pel planes live in
 *     LDS; every SAD runs out of LDS;
 *   - SADs use v_qsad_pk_u16_u8: one instruction = 4 neighbouring search positions x 4 pixels;
 *     operands are dword-aligned by construction of the LDS layouts (search position 0 sits on a
 *     dword boundary), sub-pel candidates are fetched with v_alignbyte_b32;
 *   - arg-min with the reference's "first minimum in raster order" rule is an order-independent
 *     unsigned min over keys (sad << 32 | raster_index), reduced with ds_min_u64 (LDS atomics);
 *   - integer pixel work, no MFMA.
 *
 * The code is written as a sequence of PHASES.  Inside a phase each of the 256 threads runs a
 * grid-stride loop over independent tasks; phases communicate only through LDS, separated by
 * workgroup barriers.  The uniform control flow between phases reads its inputs from LDS.  This lets
 * the identical source be compiled (a) by hipcc as the device kernel and (b) by a host compiler as a
 * serial emulation used only by the CPU test-suite to debug the kernel logic (tests/emu/).
 */
#ifndef SVT_ME_CORE_H
#define SVT_ME_CORE_H

#include <stdint.h>
#include <string.h>
#include "../../include/svtvp9_hip.h"

#ifdef SVT_HOST_EMU
#define SVT_DEV static inline
#define SVT_NT 256
static inline uint64_t svt_qsad(uint64_t ref8, uint32_t src4, uint64_t acc) {
    uint64_t out = 0;
    for (int o = 0; o < 4; o++) {
        uint32_t s = 0;
        for (int b = 0; b < 4; b++) {
            int r = (int)((ref8 >> (8 * (o + b))) & 0xff), c = (int)((src4 >> (8 * b)) & 0xff);
            s += (uint32_t)(r > c ? r - c : c - r);
        }
        out |= (uint64_t)((uint16_t)(((acc >> (16 * o)) & 0xffff) + s)) << (16 * o);
    }
    return out;
}
static inline uint32_t svt_ssd4(uint32_t a, uint32_t b, uint32_t acc) {
    for (int i = 0; i < 4; i++) {
        int d = (int)((a >> (8 * i)) & 0xff) - (int)((b >> (8 * i)) & 0xff);
        acc += (uint32_t)(d * d);
    }
    return acc;
}
static inline uint32_t svt_sad4(uint32_t a, uint32_t b, uint32_t acc) {
    for (int i = 0; i < 4; i++) {
        int x = (int)((a >> (8 * i)) & 0xff), y = (int)((b >> (8 * i)) & 0xff);
        acc += (uint32_t)(x > y ? x - y : y - x);
    }
    return acc;
}
/* per-byte (a + b + 1) >> 1 without carries between bytes */
static inline uint32_t svt_avg4(uint32_t a, uint32_t b) { return (a | b) - (((a ^ b) >> 1) & 0x7f7f7f7fu); }
static inline uint32_t svt_alignbyte(uint32_t hi, uint32_t lo, uint32_t sh) {
    return (uint32_t)(((((uint64_t)hi) << 32) | lo) >> (8 * (sh & 3)));
}
static inline void svt_lds_min_u64(uint64_t *p, uint64_t v) { if (v < *p) *p = v; }
static inline void svt_lds_add_u32(uint32_t *p, uint32_t v) { *p += v; }
/* host emulation runs lanes one after the other: a "wave reduction" degenerates to the per-lane update */
static inline void svt_wave_add_u32(uint32_t *p, uint32_t v, int uniform_dst) { (void)uniform_dst; *p += v; }
static inline void svt_wave_min_u64(uint64_t *p, uint64_t v) { if (v < *p) *p = v; }
static inline void svt_group_add_u32(uint32_t *p, uint32_t v, int group) { (void)group; *p += v; }
static inline void svt_group_add_var(uint32_t *p, uint32_t v, int group) { (void)group; *p += v; }
#define ME_MUL(a, b) ((a) * (b))
#define SVT_SCHED_FENCE() ((void)0)
/* per 16-bit lane: min(max(v, 32), 287) - 32 */
static inline uint32_t svt_pk_clamp_sub32(uint32_t v) {
    uint32_t lo = v & 0xffffu, hi = v >> 16;
    lo = (lo < 32 ? 32 : lo > 287 ? 287 : lo) - 32;
    hi = (hi < 32 ? 32 : hi > 287 ? 287 : hi) - 32;
    return lo | (hi << 16);
}
#else
#include <hip/hip_runtime.h>
#define SVT_DEV __device__ __forceinline__
#define SVT_NT 256
SVT_DEV uint64_t svt_qsad(uint64_t ref8, uint32_t src4, uint64_t acc) {
    return __builtin_amdgcn_qsad_pk_u16_u8(ref8, src4, acc);
}
SVT_DEV uint32_t svt_sad4(uint32_t a, uint32_t b, uint32_t acc) { return __builtin_amdgcn_sad_u8(a, b, acc); }
/* sum of squared differences of 4 packed samples: a.a + b.b - 2 a.b with three v_dot4_u32_u8 */
SVT_DEV uint32_t svt_ssd4(uint32_t a, uint32_t b, uint32_t acc) {
    acc = __builtin_amdgcn_udot4(a, a, acc, false);
    acc = __builtin_amdgcn_udot4(b, b, acc, false);
    return acc - 2u * __builtin_amdgcn_udot4(a, b, 0u, false);
}
SVT_DEV uint32_t svt_alignbyte(uint32_t hi, uint32_t lo, uint32_t sh) { return __builtin_amdgcn_alignbyte(hi, lo, sh); }
/* per-byte (a + b + 1) >> 1: v_lerp_u8 with the rounding bit set in every byte of the third operand */
SVT_DEV uint32_t svt_avg4(uint32_t a, uint32_t b) { return __builtin_amdgcn_lerp(a, b, 0x01010101u); }
/* per 16-bit lane: min(max(v, 32), 287) - 32 (v_pk_max_u16 / v_pk_min_u16 / v_pk_sub_u16) */
typedef unsigned short svt_u16x2 __attribute__((ext_vector_type(2)));
SVT_DEV uint32_t svt_pk_clamp_sub32(uint32_t v) {
    svt_u16x2 x = __builtin_bit_cast(svt_u16x2, v);
    const svt_u16x2 lo = {32, 32}, hi = {287, 287};
    x = __builtin_elementwise_min(__builtin_elementwise_max(x, lo), hi) - lo;
    return __builtin_bit_cast(uint32_t, x);
}
/* keeps the instruction scheduler from interleaving unrolled iterations (and their live registers) */
#define SVT_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
SVT_DEV void svt_lds_min_u64(uint64_t *p, uint64_t v) { /* lanes of one instruction must target different addresses */
    __hip_atomic_fetch_min((unsigned long long *)p, (unsigned long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
SVT_DEV void svt_lds_add_u32(uint32_t *p, uint32_t v) { atomicAdd(p, v); }
/* products of small offsets (rows, strides: far below 2^23): full-rate 24-bit multiply instead of v_mul_lo_u32 */
#define ME_MUL(a, b) __mul24((int)(a), (int)(b))
/* Cross-lane reductions use DPP row shifts (a few cycles each) instead of ds_bpermute shuffles (~90 cycles each,
 * measured), and never let several lanes of one instruction hit the same LDS address with an atomic (~100 cycles
 * per lane, measured with tools/ubench_me.hip). */
#define SVT_DPP_ADD(v, ctrl) ((v) + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), (ctrl), 0xf, 0xf, true))
/* inclusive prefix sum inside each row of 16 lanes, up to `span` lanes back (span = 4, 8 or 16) */
SVT_DEV uint32_t svt_row_prefix_add(uint32_t v, int span) {
    v = SVT_DPP_ADD(v, 0x111); /* row_shr:1 */
    v = SVT_DPP_ADD(v, 0x112); /* row_shr:2 */
    if (span >= 8) v = SVT_DPP_ADD(v, 0x114);
    if (span >= 16) v = SVT_DPP_ADD(v, 0x118);
    return v;
}
/* sum over the 64 lanes of the wave (all lanes must call; inactive contributions pass 0) then ONE LDS atomic */
SVT_DEV void svt_wave_add_u32(uint32_t *p, uint32_t v, int uniform_dst) {
    (void)uniform_dst;
    v = svt_row_prefix_add(v, 16);
    const uint32_t t = (uint32_t)__builtin_amdgcn_readlane((int)v, 15) + (uint32_t)__builtin_amdgcn_readlane((int)v, 31) +
                       (uint32_t)__builtin_amdgcn_readlane((int)v, 47) + (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
    if ((threadIdx.x & 63) == 0 && t) atomicAdd(p, t);
}
/* sum over the wave, result in every lane's *v (all lanes must call) */
SVT_DEV void svt_wave_add_u32_to_lane0(uint32_t *v) {
    uint32_t x = svt_row_prefix_add(*v, 16);
    *v = (uint32_t)__builtin_amdgcn_readlane((int)x, 15) + (uint32_t)__builtin_amdgcn_readlane((int)x, 31) +
         (uint32_t)__builtin_amdgcn_readlane((int)x, 47) + (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
}
/* min over the wave of 64-bit keys (all lanes must call; pass ~0 for "nothing"), then ONE LDS atomic */
SVT_DEV void svt_wave_min_u64(uint64_t *p, uint64_t v) {
#define SVT_DPP_MIN64(ctrl) do { \
        const uint32_t oh_ = (uint32_t)__builtin_amdgcn_update_dpp((int)(v >> 32), (int)(v >> 32), (ctrl), 0xf, 0xf, false); \
        const uint32_t ol_ = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)v, (int)(uint32_t)v, (ctrl), 0xf, 0xf, false); \
        const uint64_t w_ = ((uint64_t)oh_ << 32) | ol_; v = w_ < v ? w_ : v; } while (0)
    SVT_DPP_MIN64(0x111); SVT_DPP_MIN64(0x112); SVT_DPP_MIN64(0x114); SVT_DPP_MIN64(0x118);
#undef SVT_DPP_MIN64
    uint64_t m = ~0ull;
    _Pragma("unroll") for (int l = 15; l < 64; l += 16) {
        const uint64_t w = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(v >> 32), l) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l);
        m = w < m ? w : m;
    }
    if ((threadIdx.x & 63) == 0 && m != ~0ull) __hip_atomic_fetch_min((unsigned long long *)p, (unsigned long long)m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
/* sum over aligned groups of `group` (4, 8, 16 or 64) consecutive lanes that share one destination; whole groups are
 * active or inactive together.  The last lane of the group holds the sum and issues the LDS add. */
SVT_DEV void svt_group_add_u32(uint32_t *p, uint32_t v, int group) {
    v = svt_row_prefix_add(v, group);
    if (group == 64)
        v = (uint32_t)__builtin_amdgcn_readlane((int)v, 15) + (uint32_t)__builtin_amdgcn_readlane((int)v, 31) +
            (uint32_t)__builtin_amdgcn_readlane((int)v, 47) + (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
    if ((threadIdx.x & (group - 1)) == (unsigned)(group - 1) && v) atomicAdd(p, v);
}
/* as svt_group_add_u32, but the group size (1, 2, 4, 8 or 16 consecutive lanes, aligned) may differ from lane to lane inside a
 * wave: every lane runs the four row-shift steps and picks the partial sum that covers its own group */
SVT_DEV void svt_group_add_var(uint32_t *p, uint32_t v, int group) {
    const uint32_t s1 = SVT_DPP_ADD(v, 0x111);
    const uint32_t s2 = SVT_DPP_ADD(s1, 0x112); /* 4 lanes */
    const uint32_t s3 = SVT_DPP_ADD(s2, 0x114);
    const uint32_t s4 = SVT_DPP_ADD(s3, 0x118); /* 16 lanes */
    const uint32_t t = group == 1 ? v : group == 2 ? s1 : group == 4 ? s2 : group == 8 ? s3 : s4;
    if ((threadIdx.x & (group - 1)) == (unsigned)(group - 1) && t) atomicAdd(p, t);
}
#endif

#define ME_SB 64
#define ME_MAX_SAD_VALUE (64 * 64 * 255)

#ifdef SVT_HOST_EMU /* reference tables: the kernel derives them arithmetically, the emulation checks that */
/* raster index -> search (z-order) index, Codec/EbMotionEstimation.c:51-54 */
__attribute__((unused)) static
    const uint8_t me_tab32x32[16] = {0, 1, 4, 5, 2, 3, 6, 7, 8, 9, 12, 13, 10, 11, 14, 15};
__attribute__((unused)) static
    const uint8_t me_tab8x8[64] = {0,  1,  4,  5,  16, 17, 20, 21, 2,  3,  6,  7,  18, 19, 22, 23, 8,  9,  12, 13, 24, 25,
                                   28, 29, 10, 11, 14, 15, 26, 27, 30, 31, 32, 33, 36, 37, 48, 49, 52, 53, 34, 35, 38, 39,
                                   50, 51, 54, 55, 40, 41, 44, 45, 56, 57, 60, 61, 42, 43, 46, 47, 58, 59, 62, 63};

/* inverse maps: search (z-order) index -> raster index */
__attribute__((unused)) static
    const uint8_t me_inv32x32[16] = {0, 1, 4, 5, 2, 3, 6, 7, 8, 9, 12, 13, 10, 11, 14, 15};
__attribute__((unused)) static
    const uint8_t me_inv8x8[64] = {0,  1,  8,  9,  2,  3,  10, 11, 16, 17, 24, 25, 18, 19, 26, 27, 4,  5,  12, 13, 6,  7,
                                   14, 15, 20, 21, 28, 29, 22, 23, 30, 31, 32, 33, 40, 41, 34, 35, 42, 43, 48, 49, 56, 57,
                                   50, 51, 58, 59, 36, 37, 44, 45, 38, 39, 46, 47, 52, 53, 60, 61, 54, 55, 62, 63};
#endif

/* raster index -> search (z-order) index of the 8x8 / 16x16 PUs (Codec/EbMotionEstimation.c:51-54) by bit
 * interleaving: raster = y*8 + x (3+3 bits) or y*4 + x (2+2 bits), z = ... y1 x1 y0 x0 */
SVT_DEV int me_z8(int b) { return (b & 1) | ((b & 2) << 1) | ((b & 4) << 2) | ((b & 8) >> 2) | (b & 16) >> 1 | (b & 32); }
SVT_DEV int me_z4(int b) { return (b & 1) | ((b & 2) << 1) | ((b & 4) >> 1) | (b & 8); }

/* The sub-pel candidate tables are packed into immediates so that no phase has to fetch them from memory:
 * a nibble holds plane (2 bits) | dx flag << 2 | dy flag << 3 (flag = -1 for the search tables, +1 for bi-pred). */
#define ME_HCAND_PACK 0x73BF2A15u
SVT_DEV void me_hcand_get(int cand, int *plane, int *dx, int *dy) {
    uint32_t n = (ME_HCAND_PACK >> (4 * cand)) & 15u;
    *plane = (int)(n & 3); *dx = -(int)((n >> 2) & 1); *dy = -(int)(n >> 3);
}
SVT_DEV uint32_t me_qtab_get(int method, int pos) { /* byte: first source nibble | second source nibble << 4 */
    const uint64_t v = method == 0 ? 0x25121AA5200A1005ull : method == 1 ? 0x5625A55E755F0554ull
                     : method == 2 ? 0xA51A9AAD0AA8BAAFull : 0x5EA5ADDE5FFDAFFEull;
    return (uint32_t)(v >> (8 * pos)) & 0xffu;
}
SVT_DEV uint32_t me_btab_get(int frac, int *has_b) {
    const uint64_t v = frac < 8 ? 0x6131212041011000ull : 0x9693928263033202ull;
    *has_b = (int)((0xFAFAu >> frac) & 1u);
    return (uint32_t)(v >> (8 * (frac & 7))) & 0xffu;
}
/* sign of the candidate displacement (L,R,T,B,TL,TR,BR,BL): half-pel moves by 2, quarter-pel by 1 quarter sample */
SVT_DEV void me_dmv_get(int i, int *sx, int *sy) {
    uint32_t n = (0x8A209164u >> (4 * i)) & 15u;
    *sx = (int)(n & 3) - 1; *sy = (int)(n >> 2) - 1;
}

/* Picture descriptor as seen by the kernel (device pointers inside the planes). */
typedef struct me_pic_dev {
    svt_pa_picture    cur, ref[2];
    svt_me_pu_result *results;
    uint32_t         *rcme;
    /* the parameters that change from picture to picture inside a configuration (me_spec.h) and what the host derives from
     * them (HME level-0 areas scaled by the temporal layer's multiplier): one launch serves pictures of several layers */
    uint8_t           num_ref_lists, temporal_layer_index, hierarchical_levels, same_ref_poc;
    int16_t           hme_w0[2], hme_h0[2], hme_tw0, hme_th0;
    int16_t           hme_band; /* me_fast.h: search rows of the (widest) level-0 window that fit the LDS scratch at a time */
} me_pic_dev;

/* LDS layout (byte offsets), computed on the host from the parameters (me_lds_layout) */
typedef struct me_lds_layout {
    int32_t off_state;   /* me_state_t */
    int32_t off_src;     /* 64 x 64 source SB, stride 64 */
    int32_t off_region;  /* integer reference samples of the current list's search region */
    int32_t off_planes;  /* B, H, J half-pel planes (3 x plane_bytes); aliased by HME window / SAD scratch */
    int32_t off_quarter; /* 32x32 quarter-resolution SB (only when HME level 1 is enabled) */
    int32_t off_ssd;     /* SSD_SEARCH only: candidate SSDs [85][9]; entry 8 of a PU is the integer position's, then the PU's best so far */
    int32_t off_cand;    /* sub-pel candidate distortions [pu][8] / bi-pred distortion [pu]: entries 0..167 (PUs 0..20), dwords */
    int32_t off_cand_hi; /* entries 168..679 (the 8x8 PUs: at most 2 x 64 x 255 each) as halfwords, when cand_dwords = 680; else -1 */
    int32_t cand_dwords; /* 8 x (21 when the 8x8 PUs are never refined nor bi-predicted, else 85) */
    int32_t off_pred0;   /* host emulation only (the kernel keeps them in registers): list 0 prediction of the bi-pred lanes */
    int32_t region_stride, region_rows;
    int32_t plane_stride; /* row stride of the half-pel planes: they are narrower than the region (no search tail) */
    int32_t plane_bytes;
    int32_t scratch_bytes; /* bytes available at off_planes */
    int32_t total_bytes;
    int32_t compact;     /* me_layout.h: no tail columns in the region rows, quarter SB inside the SSD tables */
    /* HME level-0 search areas already scaled by the temporal layer's multiplier (Codec/EbDefinitions.h:989-1005): the
     * divisions by 100 are done once per launch on the host instead of by the planning thread of every SB */
    int16_t hme_w0[2], hme_h0[2], hme_tw0, hme_th0;
} me_lds_layout;

#define ME_RGN_GX 4 /* left guard columns of the region buffer (search position 0 is dword aligned) */
#define ME_RGN_GY 3 /* top guard rows */
#define ME_PL_G 2   /* guard of the half-pel planes */

/* ---- HME work list: (region, band of search rows) windows staged in the scratch and searched batch by batch ---- */
#define ME_HME_MAX_WIN 16
typedef struct me_hme_win {
    int16_t  gx, gy;         /* reference-picture coordinates of window column 0 / row 0 */
    uint32_t off;            /* byte offset of the window inside the scratch (the scratch can exceed 64 KB: search areas up to 127 x 127) */
    uint16_t wstride;        /* window row stride (bytes, odd number of dwords) */
    uint16_t tl, ts;         /* first load task / first search task of this window inside its batch */
    uint16_t sw, sh;         /* search positions */
    uint16_t y0;             /* first search row of this window inside its region (row band offset) */
    uint16_t rows;           /* window rows */
    uint8_t  nd;             /* window dwords per row */
    uint8_t  slot;           /* region (key) this window belongs to */
    uint32_t inv_nu, inv_ng; /* me_magic_of(16-byte units per window row) / (search tasks per search row): the planning thread divides once */
} me_hme_win;

/* per-SB state in LDS */
typedef struct me_state_t {
    union {
        uint64_t key[85];      /* full-pel arg-min keys of the current list */
        struct {               /* HME work list: dead once the level's results are in hme_x/y/sad, before the keys are set */
            int32_t    hme_nbatch, hme_bstart[ME_HME_MAX_WIN + 1]; /* batches of windows that fit the scratch together */
            me_hme_win hme_win[ME_HME_MAX_WIN];
        };
    };
    uint64_t hme_key;          /* arg-min key of the stand-alone SAD-loop kernel */
    uint64_t hme_keys[4];      /* arg-min keys of the region searches of the current HME level */
    uint64_t hme_sad[3][4];    /* per level, per region slot (rh*2 + rw): best SAD * 2 */
    int16_t  hme_x[3][4], hme_y[3][4]; /* per level, per region slot: search centre in / best position out */
    int16_t  hme_cox[4], hme_coy[4], hme_cw[4], hme_ch[4]; /* clipped search areas of the current level */
    int16_t  hme_xc, hme_yc;   /* HME result; persists from list 0 to list 1 when no level runs */
    int32_t  hme_rh;           /* [quirk] the reference's region-row counter, not reset between the lists */
    uint32_t best_sad[2][85];  /* search (z-order) index */
    uint32_t best_mv[2][85];
    uint32_t red[8];           /* small sum reductions */
    uint32_t spu[85];          /* refined PUs of the current list, dense: pu | n << 7 | (px>>3) << 14 | (py>>3) << 17 | log2(w/8) << 20 */
    uint32_t supel[9];         /* su_pel_enable sums: sx,sy,ssad for 32/16/8 */
    svt_plane refd[3];         /* descriptors (full, 1/4, 1/16) of the current list's reference picture, copied from HBM once */
    uint8_t  dir[88];          /* 85 used; padded so that the block below stays dword aligned */
    /* rows 0,2,4.. of the 1/16-resolution SB, read as dwords by the HME search: a misaligned ds_read is replayed at ~64
     * cycles per wave-instruction (SQ_LDS_UNALIGNED_STALL was 2/3 of all LDS cycles of the kernel before this was aligned) */
    uint8_t  sixteenth_sb[16 * 8] __attribute__((aligned(16)));
} me_state_t;

SVT_DEV int16_t me_mvx(uint32_t mv) { return (int16_t)(mv & 0xFFFF); }
SVT_DEV int16_t me_mvy(uint32_t mv) { return (int16_t)(mv >> 16); }
SVT_DEV uint32_t me_pack_mv(int x, int y) { return ((uint32_t)(uint16_t)y << 16) | (uint16_t)x; }
SVT_DEV const uint8_t *me_pix(const svt_plane *p, int x, int y) {
    return p->buf + (ptrdiff_t)(p->origin_y + y) * p->stride + p->origin_x + x;
}
SVT_DEV int me_pu_nidx(int pu) { return pu > 20 ? me_z8(pu - 21) + 21 : pu > 4 ? me_z4(pu - 5) + 5 : pu; }
SVT_DEV void me_pu_geom(int pu, int *x, int *y, int *w) {
    if (pu == 0) { *x = 0; *y = 0; *w = 64; }
    else if (pu < 5) { *x = ((pu - 1) & 1) * 32; *y = ((pu - 1) >> 1) * 32; *w = 32; }
    else if (pu < 21) { *x = ((pu - 5) & 3) * 16; *y = ((pu - 5) >> 2) * 16; *w = 16; }
    else { *x = ((pu - 21) & 7) * 8; *y = ((pu - 21) >> 3) * 8; *w = 8; }
}

/* unaligned 32-bit fetch from a byte address (LDS or global): two aligned loads + v_alignbyte.  Branch-free on
 * purpose: a conditional second load serialises the two memory round trips and keeps the compiler from batching the
 * loads of unrolled callers.  An aligned address re-reads its own dword, so nothing beyond the 4 bytes is touched. */
/* Pointers into picture planes and result arrays are known to be global memory: saying so turns the generic
 * (flat_load: 64-bit address per lane, aperture check, counted against the LDS counter too) accesses into global_load /
 * global_store, which also accept a scalar base plus a 32-bit lane offset. */
#ifdef SVT_HOST_EMU
#define SVT_GLOBAL
#else
#define SVT_GLOBAL __attribute__((address_space(1)))
#endif
#define SVT_AS_GLOBAL(T, p) ((T SVT_GLOBAL *)(uintptr_t)(p))
/* 32 bits from a global byte address of any alignment: ONE load.  Global (and scratch) accesses need no alignment on this
 * target (the compiler emits a single global_load_dword for an align-1 dword; the texture unit splits the rare access that
 * straddles a line) -- only LDS penalises misalignment, which is why me_ld32u below still assembles its dword from two. */
#ifdef SVT_HOST_EMU
SVT_DEV uint32_t me_ld32u_g(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
#else
typedef uint32_t __attribute__((aligned(1))) me_u32_unaligned;
SVT_DEV uint32_t me_ld32u_g(const uint8_t *p) { return *SVT_AS_GLOBAL(const me_u32_unaligned, p); }
#endif
/* 8 / 16 bytes from a global byte address of any alignment: one global_load_dwordx2 / x4 */
typedef struct me_u32x2 { uint32_t x, y; } me_u32x2;
typedef struct me_u32x4 { uint32_t x, y, z, w; } me_u32x4;
#ifdef SVT_HOST_EMU
SVT_DEV me_u32x2 me_ld64u_g(const uint8_t *p) { me_u32x2 v; memcpy(&v, p, 8); return v; }
SVT_DEV me_u32x4 me_ld128u_g(const uint8_t *p) { me_u32x4 v; memcpy(&v, p, 16); return v; }
#else
typedef uint32_t me_v2u __attribute__((ext_vector_type(2), aligned(1)));
typedef uint32_t me_v4u __attribute__((ext_vector_type(4), aligned(1)));
SVT_DEV me_u32x2 me_ld64u_g(const uint8_t *p) { const me_v2u t = *SVT_AS_GLOBAL(const me_v2u, p); me_u32x2 v = {t.x, t.y}; return v; }
SVT_DEV me_u32x4 me_ld128u_g(const uint8_t *p) { const me_v4u t = *SVT_AS_GLOBAL(const me_v4u, p); me_u32x4 v = {t.x, t.y, t.z, t.w}; return v; }
#endif
SVT_DEV uint32_t me_ld32u(const uint8_t *p) {
    const uint32_t  sh = (uint32_t)((uintptr_t)p & 3);
    const uint32_t *q  = (const uint32_t *)(p - sh);
    const uint32_t  lo = q[0], hi = q[sh ? 1 : 0];
    return svt_alignbyte(hi, lo, sh);
}

/* [quirk] origin is updated first and the width test re-evaluated afterwards, so left/top clipping never
 * shrinks the area (Codec/EbMotionEstimation.c:5022-5054 and the HME copies). */
SVT_DEV void me_clip_area(int origin, int16_t *area_origin, int16_t *area_size, int pad, int pic_dim) {
    int16_t o = *area_origin, s = *area_size;
    o = (int16_t)(((origin + o) < -pad) ? -pad - origin : o);
    s = (int16_t)(((origin + o) < -pad) ? s - (-pad - (origin + o)) : s);
    o = (int16_t)(((origin + o) > pic_dim - 1) ? o - ((origin + o) - (pic_dim - 1)) : o);
    if ((origin + o + s) > pic_dim) {
        int t = s - ((origin + o + s) - pic_dim);
        s     = (int16_t)(t > 1 ? t : 1);
    }
    *area_origin = o;
    *area_size   = s;
}
SVT_DEV int16_t me_clip_center(int origin, int16_t c, int pad, int pic_dim) {
    c = (int16_t)(((origin + c) < -pad) ? -pad - origin : c);
    c = (int16_t)(((origin + c) > pic_dim - 1) ? c - ((origin + c) - (pic_dim - 1)) : c);
    return c;
}

/* A rectangle of global memory addressed as one uniform base (scalar registers -> the loads use the scalar-base addressing
 * form with a 32-bit lane offset, no 64-bit address arithmetic per lane) plus byte offsets. */
typedef struct me_gsrc { const uint8_t *base; } me_gsrc;
SVT_DEV me_gsrc me_gsrc_of(const uint8_t *p) {
    me_gsrc   g;
    uintptr_t a = (uintptr_t)p;
#ifndef SVT_HOST_EMU
    a  = ((uintptr_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a);
#endif
    g.base = (const uint8_t *)a;
    return g;
}
/* the 4 bytes at byte offset off of the rectangle */
SVT_DEV uint32_t me_gld(const me_gsrc g, uint32_t off) { return me_ld32u_g(g.base + off); }

/* a plane descriptor read from LDS (or HBM) into scalar registers: every lane holds the same values, and with them in
 * SGPRs the address arithmetic built on them (clipping, me_pix, row offsets) runs on the scalar unit */
SVT_DEV svt_plane me_plane_uni(const svt_plane *p) {
    svt_plane u;
#ifdef SVT_HOST_EMU
    u = *p;
#else
    const uintptr_t a = (uintptr_t)p->buf;
    u.buf = (const uint8_t *)(((uintptr_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32)) << 32) |
                              (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a));
    u.stride = __builtin_amdgcn_readfirstlane(p->stride); u.origin_x = __builtin_amdgcn_readfirstlane(p->origin_x);
    u.origin_y = __builtin_amdgcn_readfirstlane(p->origin_y); u.width = __builtin_amdgcn_readfirstlane(p->width);
    u.height = __builtin_amdgcn_readfirstlane(p->height);
#endif
    return u;
}

/* two groups per iteration in the fused full-pel phase: the configurations with a 64x64 search area (few waves per SIMD) */
/* two groups per iteration of the fused full-pel loop (independent chains) for the large areas; four were measured on the 64x64 area
 * of C5 and gain nothing (the phase runs at its issue rate there: 44 % of the workgroup's time either way) */
#define ME_FULLPEL_UNROLL2(c) ((c)->p->search_area_width * (c)->p->search_area_height >= 2048)

/* everything a phase needs */
typedef struct me_ctx_t {
    const me_pic_dev    *pic;
    const svt_me_params *p;
    me_lds_layout        L;
    uint8_t             *lds;
    me_state_t          *st;
    uint8_t             *src;    /* LDS */
    uint8_t             *region; /* LDS */
    uint8_t             *planes; /* LDS */
    uint8_t             *hme_scratch; /* LDS: where the HME levels stage their windows -- the region buffer and the planes behind it, both dead while a list's
                                         hierarchical search runs (the region is staged after it, the planes are interpolated from the region) */
    int                  hme_scratch_bytes;
    uint8_t             *quarter_sb; /* LDS, valid when HME level 1 is enabled */
    uint32_t            *ssdc;       /* LDS, SSD_SEARCH only: SSD of the sub-pel candidates [pu][9] (8 = integer position) */
    uint32_t            *cand;       /* LDS: sub-pel candidate distortions [pu][8] (see me_cand_get); bi-pred distortion [pu] */
    uint32_t            *cand_hi;    /* LDS: the 8x8 PUs' entries of that table as halfwords (L.off_cand_hi >= 0: else they are never refined, and this is cand) */
    uint32_t            *pred0;  /* host emulation only: list 0 prediction dwords of the bi-pred lanes [16][256] */
    int                  pic_w, pic_h, sb_x, sb_y, sb_w, sb_h, sb_index;
    unsigned long long  *prof;   /* optional per-phase cycle accumulators (profiling builds), else NULL */
    uint32_t            *redo;   /* compact layout: set to 1 when this SB needs the full layout (its clipped search area has tail columns); else NULL */
} me_ctx_t;

/* t / d for a small wave-uniform divisor d (phase geometry: units per row, lanes per strip, search width ...).  An integer
 * division costs ~25 vector instructions per wave here; the reciprocals of 1..256 sit in constant memory instead (one scalar
 * load) and the quotient is one v_mul_hi: exact while t * d < 2^32 (inv = floor((2^32 - 1) / d) + 1; d = 1 -> inv = 0 -> t). */
#ifdef SVT_HOST_EMU
static inline int me_udiv(int t, int d) { return t / d; }
#else
struct me_magic_table {
    uint32_t v[257];
    constexpr me_magic_table() : v() { for (uint32_t d = 1; d <= 256; d++) v[d] = (uint32_t)(0xffffffffu / d) + 1u; }
};
__constant__ const me_magic_table me_magics = me_magic_table();
SVT_DEV int me_udiv(int t, int d) {
    const int du = __builtin_amdgcn_readfirstlane(d);
    if (du > 256) return t / du;
    const uint32_t inv = me_magics.v[du];
    return inv ? (int)__umulhi((uint32_t)t, inv) : t;
}
#endif

/* ------------------------------------------------------------------------------------------------ */
/* phases (each: grid-stride loop over tasks; tid in [0,256))                                         */
/* ------------------------------------------------------------------------------------------------ */

/* copy a w_bytes x rows rectangle from global memory (any alignment) into LDS (dst 4-byte aligned rows) */
SVT_DEV void ph_load_rect(int tid, uint8_t *dst, int dst_stride, const uint8_t *src, int src_stride, int w_bytes, int rows) {
    /* task = 16 bytes of a row: one global load (four times fewer memory instructions than dwords -- the texture unit's
     * instruction rate, not bandwidth, is what these small rectangles cost), four LDS dword stores; the last unit of a row
     * is shortened to whole dwords so that nothing beyond the rectangle's last dword is read or written.  A thread's
     * (row, unit) pair advances by SVT_NT tasks per step without a division; the loads of two steps are issued before the
     * first LDS store. */
    const int nd = (w_bytes + 3) >> 2, nu = (nd + 3) >> 2, n = nu * rows;
    const int dr = me_udiv(SVT_NT, nu), di = SVT_NT - dr * nu;
    int       r = me_udiv(tid, nu), i = tid - r * nu;
    const me_gsrc g = me_gsrc_of(src);
    for (int t0 = tid; t0 < n; t0 += 2 * SVT_NT) {
        me_u32x4 v[2];
        int      o[2], k[2];
        _Pragma("unroll") for (int u = 0; u < 2; u++) {
            o[u] = -1; k[u] = 0;
            if (t0 + u * SVT_NT < n) {
                const uint32_t goff = (uint32_t)(ME_MUL(r, src_stride) + 16 * i);
                k[u] = nd - 4 * i < 4 ? nd - 4 * i : 4; /* dwords of this unit */
                if (k[u] == 4) v[u] = me_ld128u_g(g.base + goff);
                else {
                    v[u].x = me_gld(g, goff);
                    v[u].y = k[u] > 1 ? me_gld(g, goff + 4) : 0;
                    v[u].z = k[u] > 2 ? me_gld(g, goff + 8) : 0;
                    v[u].w = 0;
                }
                o[u] = r * dst_stride + 16 * i;
            }
            i += di; r += dr;
            if (i >= nu) { i -= nu; r++; }
        }
        _Pragma("unroll") for (int u = 0; u < 2; u++)
            if (o[u] >= 0) {
                uint32_t *d = (uint32_t *)(dst + o[u]);
                d[0] = v[u].x;
                if (k[u] > 1) d[1] = v[u].y;
                if (k[u] > 2) d[2] = v[u].z;
                if (k[u] > 3) d[3] = v[u].w;
            }
    }
}

/* the 1/4-resolution SB, stride 32 */
SVT_DEV void ph_load_quarter(const me_ctx_t *c, int tid) {
    int rows = c->sb_h >> 1, wq = c->sb_w >> 1;
    for (int t = tid; t < rows * 32; t += SVT_NT) {
        int r = t >> 5, x = t & 31;
        c->quarter_sb[t] = x < wq ? *SVT_AS_GLOBAL(const uint8_t, me_pix(&c->pic->cur.quarter, (c->sb_x >> 1) + x, (c->sb_y >> 1) + r)) : 0;
    }
}

/* initial state + decimated SB copies (Codec/EbMotionEstimationProcess.c:984-1035) */
SVT_DEV void ph_init(const me_ctx_t *c, int tid) {
    me_state_t *st = c->st;
    for (int t = tid; t < 85; t += SVT_NT) {
        st->best_mv[0][t] = 0; st->best_mv[1][t] = 0;
        st->best_sad[0][t] = 0; st->best_sad[1][t] = 0;
        st->dir[t] = 0;
    }
    if (tid < 8) st->red[tid] = 0;
    if (tid < 12) { st->hme_x[tid >> 2][tid & 3] = 0; st->hme_y[tid >> 2][tid & 3] = 0; st->hme_sad[tid >> 2][tid & 3] = 0; }
    if (tid == 0) { st->hme_rh = 0; st->hme_xc = 0; st->hme_yc = 0; }
    /* source SB: always 64x64 from the padded picture */
    ph_load_rect(tid, c->src, ME_SB, me_pix(&c->pic->cur.full, c->sb_x, c->sb_y), c->pic->cur.full.stride, ME_SB, ME_SB);
    if (c->p->enable_hme_level_0_flag) {
        /* rows 0,2,4,.. of the 1/16 SB, stride 16 */
        int rows = (c->sb_h >> 2) >> 1, wq = c->sb_w >> 2;
        for (int t = tid; t < rows * 16; t += SVT_NT) {
            int r = t >> 4, x = t & 15;
            st->sixteenth_sb[t] = x < wq ? *SVT_AS_GLOBAL(const uint8_t, me_pix(&c->pic->cur.sixteenth, (c->sb_x >> 2) + x, (c->sb_y >> 2) + 2 * r)) : 0;
        }
    }
    if (c->p->enable_hme_level_1_flag) ph_load_quarter(c, tid);
}

/* Row-subsampled 64-wide SADs of the source SB against up to 5 displaced reference blocks read from
 * global memory (test_search_area_bounds / check_zero_zero_center).  Each thread: one (row, 8-byte) piece.
 * Results accumulate in st->red[k]; caller doubles them. */
SVT_DEV void ph_center_sads(const me_ctx_t *c, int tid, const svt_plane *ref, int ncand, const int16_t *dx, const int16_t *dy) {
    const int rows = c->sb_h >> 1, wq = c->sb_w >> 3; /* 8-byte pieces per row (SB widths are multiples of 8) */
    const int n = rows * wq;                          /* <= 256: one piece per thread */
    uint32_t  acc[5] = {0, 0, 0, 0, 0};
    me_u32x2  v[5], s = {0, 0};
    /* every thread takes part in the wave reductions below; the (independent) global loads of all candidates are issued before
     * the first use: one memory round trip for the phase, one 8-byte load per candidate and thread */
    _Pragma("unroll") for (int k = 0; k < 5; k++) v[k] = s;
    if (tid < n) {
        const int r = me_udiv(tid, wq), i = tid - r * wq;
        const uint32_t *sp = (const uint32_t *)(c->src + (2 * r) * ME_SB + 8 * i);
        s.x = sp[0]; s.y = sp[1];
        const int rstride = ref->stride;
        _Pragma("unroll") for (int k = 0; k < 5; k++) {
            if (k < ncand) {
                const me_gsrc g = me_gsrc_of(me_pix(ref, c->sb_x + dx[k], c->sb_y + dy[k]));
                v[k] = me_ld64u_g(g.base + (uint32_t)(ME_MUL(2 * r, rstride) + 8 * i));
            } else v[k] = s;
        }
    }
    _Pragma("unroll") for (int k = 0; k < 5; k++)
        if (k < ncand) acc[k] = svt_sad4(v[k].y, s.y, svt_sad4(v[k].x, s.x, 0));
    _Pragma("unroll") for (int k = 0; k < 5; k++)
        if (k < ncand) svt_wave_add_u32(&c->st->red[k], acc[k], 1);
}

/* the same row-subsampled SAD for ONE displaced block that already sits in the LDS search region (region byte
 * (col, row) = its top-left sample); result accumulates in st->red[1] */
SVT_DEV void ph_region_center_sad(const me_ctx_t *c, int tid, int col, int row) {
    const int rows = c->sb_h >> 1, wd = c->sb_w >> 2, n = rows * wd, rs = c->L.region_stride;
    uint32_t  acc = 0;
    _Pragma("unroll") for (int h = 0; h < 2; h++) {
        const int t = tid + h * SVT_NT;
        if (t < n) {
            const int r = me_udiv(t, wd), i = t - r * wd;
            acc = svt_sad4(me_ld32u(c->region + ME_MUL(row + 2 * r, rs) + col + 4 * i), *(const uint32_t *)(c->src + (2 * r) * ME_SB + 4 * i), acc);
        }
    }
    svt_wave_add_u32(&c->st->red[1], acc, 1);
}

/* Generic exhaustive SAD search (= eb_vp9_sad_loop_kernel) over a window staged in LDS.
 * blk: block rows (already subsampled) in LDS, stride bstride, bw x bh.  win: LDS window whose row r holds
 * reference row (window_top + r) and column 0 = search x position 0; a search row y uses window rows
 * y + mul*j (j = block row; mul = 2 in every reference use).  Key = (sad << 32) | (y * sw + x).  bw multiple of 4 uses QSAD. */
SVT_DEV void ph_sad_search(const me_ctx_t *c, int tid, const uint8_t *blk, int bstride, int bw, int bh, const uint8_t *win,
                           int wstride, int sw, int sh, int mul) {
    int      ng   = (sw + 3) >> 2;
    uint64_t best = ~0ull;
    if ((bw & 3) == 0) {
        int nd = bw >> 2;
        for (int t = tid; t < ng * sh; t += SVT_NT) {
            int      y = t / ng, g = t - y * ng;
            uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0;
            for (int j = 0; j < bh; j++) {
                const uint32_t *wr  = (const uint32_t *)(win + (y + mul * j) * wstride + 4 * g);
                const uint32_t *br  = (const uint32_t *)(blk + j * bstride);
                uint64_t        acc = 0;
                uint32_t        lo  = wr[0];
                for (int i = 0; i < nd; i++) {
                    uint32_t hi = wr[i + 1];
                    acc         = svt_qsad(((uint64_t)hi << 32) | lo, br[i], acc);
                    lo          = hi;
                }
                a0 += (uint32_t)(acc & 0xffff); a1 += (uint32_t)((acc >> 16) & 0xffff);
                a2 += (uint32_t)((acc >> 32) & 0xffff); a3 += (uint32_t)(acc >> 48);
            }
            uint32_t a[4] = {a0, a1, a2, a3};
            for (int o = 0; o < 4; o++) {
                int x = 4 * g + o;
                if (x < sw) {
                    uint64_t k = ((uint64_t)a[o] << 32) | (uint32_t)(y * sw + x);
                    if (k < best) best = k;
                }
            }
        }
    } else {
        for (int t = tid; t < sw * sh; t += SVT_NT) {
            int      y = t / sw, x = t - y * sw;
            uint32_t s = 0;
            for (int j = 0; j < bh; j++)
                for (int i = 0; i < bw; i++) {
                    int a = blk[j * bstride + i], b = win[(y + mul * j) * wstride + x + i];
                    s += (uint32_t)(a > b ? a - b : b - a);
                }
            uint64_t k = ((uint64_t)s << 32) | (uint32_t)t;
            if (k < best) best = k;
        }
    }
    svt_wave_min_u64(&c->st->hme_key, best);
}

/* full-pel search tables.  All 85 PU SADs of a search position live in one row of ME_PU_STRIDE dwords indexed by
 * the PU's search-order index (0 = 64x64, 1..4 = 32x32, 5..20 = 16x16, 21..84 = 8x8; children of a block are the 4
 * consecutive entries 4*z .. 4*z+3 of the next level, i.e. nested z-order).  Entries 0..20 are dwords; the 64 8x8 SADs
 * (sub-sampled, < 2^16) follow as halfwords.  The odd stride keeps the per-position rows on different LDS banks. */
#define ME_PU_STRIDE 53

/* full-pel: sub-sampled 8x8 SADs of every (position, 8x8 block) of a chunk of search rows.
 * Task = (8x8 block b in raster order, 4-position group g, search row y).  Output U[pos][21 + z(b)]
 * (pos = y_local * sw + x).  tail columns (x >= w8) reproduce the reference's address bug
 * for 16x16 blocks 12 and 13 (Codec/EbMotionEstimation.c:855-856). */
SVT_DEV void ph_fullpel_sad8(const me_ctx_t *c, int tid, uint32_t *U, int sw, int y0, int ny, int w8) {
    int ng = (sw + 3) >> 2;
    int rs = c->L.region_stride;
    for (int t = tid; t < ng * ny * 64; t += SVT_NT) {
        int b = t & 63, q = t >> 6;
        int yl = q / ng, g = q - yl * ng;
        int bx = (b & 7) * 8, by = (b >> 3) * 8;
        int rbx = bx;
        if (4 * g >= w8) {
            /* 16x16 block (raster) containing b: z-order 12 -> raster 10 (x=32,y=32), 13 -> raster 11 (x=48,y=32) */
            if (by >= 32 && by < 48 && bx >= 32) rbx += 16;
        }
        const uint8_t *rp = c->region + ME_MUL(ME_RGN_GY + y0 + yl + by, rs) + ME_RGN_GX + 4 * g + rbx;
        const uint8_t *sp = c->src + by * ME_SB + bx;
        uint64_t       acc = 0;
        _Pragma("unroll") for (int r = 0; r < 4; r++) {
            const uint32_t *w = (const uint32_t *)(rp + 2 * r * rs);
            const uint32_t *s = (const uint32_t *)(sp + 2 * r * ME_SB);
            uint32_t        d0 = w[0], d1 = w[1], d2 = w[2];
            acc = svt_qsad(((uint64_t)d1 << 32) | d0, s[0], acc);
            acc = svt_qsad(((uint64_t)d2 << 32) | d1, s[1], acc);
        }
        uint16_t *u = (uint16_t *)(U + ME_MUL(ME_MUL(yl, sw) + 4 * g, ME_PU_STRIDE) + 21) + me_z8(b);
        _Pragma("unroll") for (int o = 0; o < 4; o++)
            if (4 * g + o < sw) u[o * 2 * ME_PU_STRIDE] = (uint16_t)(acc >> (16 * o));
    }
}

#ifndef SVT_HOST_EMU
/* The device form of ph_fullpel_fused (below).  The instruction stream of a group of 4 positions is written out: both dwords of
 * every QSAD operand are read as a pair (two ds_read2 per row instead of register moves; the lane's LDS offsets are opaque to the
 * compiler so that a group costs ONE add per operand stream and the rows are immediate offsets), a key is one v_lshl_or /
 * v_and_or and five keys meet in two v_min3, the 32x32 step adds 16-bit halves across the row without unpacking them first, and
 * the 64x64 step is eight in-place DPP adds: 56 vector instructions per group (87 before).  NG = 2 evaluates two groups per
 * iteration with independent accumulators: for the configurations whose LDS need leaves one or two waves per SIMD (64x64 search
 * areas) the phase is bound by the latency of its dependent chains, not by issue. */
typedef uint64_t __attribute__((aligned(4))) me_u64a4; /* a dword pair in LDS: ds_read2_b32 */
SVT_DEV uint32_t me_min3(uint32_t a, uint32_t b, uint32_t c) { const uint32_t m = a < b ? a : b; return m < c ? m : c; }
/* RUN (with NG = 2): the two groups of an iteration are NEIGHBOURS in a search row -- a run of 8 positions -- and share their operand pairs:
 * group 0 takes the pairs at +0 and +4 of a window row, group 1 those at +4 and +8: three reads per row instead of four (the LDS, shared by
 * the CU's five workgroups, is as busy as the vector unit in this kernel) */
template <int NG, bool RUN = false> SVT_DEV void me_fullpel_fused_dev(const me_ctx_t *c, int tid, int sw, int sh) {
    const int rs = c->L.region_stride;
    const int z = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bx = ((z & 1) | ((z >> 1) & 2) | ((z >> 2) & 4)) * 8, by = (((z >> 1) & 1) | ((z >> 2) & 2) | ((z >> 3) & 4)) * 8;
    uint32_t  s0[4], s1[4]; /* rows 0, 2, 4, 6 of the source block */
    _Pragma("unroll") for (int r = 0; r < 4; r++) {
        const uint32_t *s = (const uint32_t *)(c->src + (by + 2 * r) * ME_SB + bx);
        s0[r] = s[0]; s1[r] = s[1];
    }
    uint32_t ro0 = (uint32_t)(c->region - c->lds) + (uint32_t)(ME_MUL(ME_RGN_GY + by, rs) + ME_RGN_GX + bx), ro1 = ro0 + 4;
    __asm__("" : "+v"(ro0));
    __asm__("" : "+v"(ro1));
    uint32_t mhi = 0xffff0000u;
    __asm__("" : "+v"(mhi)); /* in a vector register: (x & mhi) | s is then ONE v_and_or_b32 (one scalar operand per instruction) */
    const int ng = RUN ? sw >> 3 : sw >> 2; /* RUN: runs per search row */
    static_assert(!RUN || NG == 2, "a run is two groups");
    uint32_t  b8 = 0xffffffffu, b16 = 0xffffffffu, b32 = 0xffffffffu, b64 = 0xffffffffu;
#define FP_DPP(v, ctrl) ((v) + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), (ctrl), 0xf, 0xf, false))
#define FP_KEYS(b, lo, hi, pos) do { \
        b = me_min3(b, ((lo) << 16) | (pos), ((lo) & mhi) | ((pos) + 1)); \
        b = me_min3(b, ((hi) << 16) | ((pos) + 2), ((hi) & mhi) | ((pos) + 3)); } while (0)
    /* group q = y * ng + g (wave-uniform; y by reciprocal multiplication on the scalar unit: the body stays one basic block); the
     * waves take the groups round-robin, NG consecutive rounds per iteration (a group past the end repeats the last one: the
     * minima do not change) */
    const uint32_t inv = me_magics.v[ng]; /* ng in [2, 31] */
    const int      nq = ME_MUL(ng, sh);
    for (int q0 = w; q0 < nq; q0 += RUN ? 4 : 4 * NG) {
        uint32_t pos[NG], lo[NG], hi[NG], a0[NG], a1[NG], a2[NG], a3[NG];
        if constexpr (RUN) {
            const int      y = inv ? (int)(((uint64_t)(uint32_t)q0 * inv) >> 32) : q0, g = 2 * (q0 - y * ng);
            const int      off = ME_MUL(y, rs) + 4 * g; /* wave-uniform */
            const uint8_t *rp = c->lds + (ro0 + (uint32_t)off), *rp1 = c->lds + (ro1 + (uint32_t)off);
            uint64_t       acc = 0, acc_b = 0;
            _Pragma("unroll") for (int r = 0; r < 4; r++) {
                const uint64_t p0 = *(const me_u64a4 *)(rp + 2 * r * rs), p1 = *(const me_u64a4 *)(rp1 + 2 * r * rs), p2 = *(const me_u64a4 *)(rp + 2 * r * rs + 8);
                acc = svt_qsad(p0, s0[r], acc);     acc = svt_qsad(p1, s1[r], acc);
                acc_b = svt_qsad(p1, s0[r], acc_b); acc_b = svt_qsad(p2, s1[r], acc_b);
            }
            pos[0] = (uint32_t)(ME_MUL(y, sw) + 4 * g); pos[NG - 1] = pos[0] + 4;
            lo[0] = (uint32_t)acc; hi[0] = (uint32_t)(acc >> 32); lo[NG - 1] = (uint32_t)acc_b; hi[NG - 1] = (uint32_t)(acc_b >> 32);
        } else
        _Pragma("unroll") for (int u = 0; u < NG; u++) {
            const int      q = q0 + 4 * u < nq ? q0 + 4 * u : q0;
            const int      y = (int)(((uint64_t)(uint32_t)q * inv) >> 32), g = q - y * ng;
            const int      off = ME_MUL(y, rs) + 4 * g; /* wave-uniform */
            const uint8_t *rp = c->lds + (ro0 + (uint32_t)off), *rp1 = c->lds + (ro1 + (uint32_t)off);
            uint64_t       acc = 0;
            _Pragma("unroll") for (int r = 0; r < 4; r++) {
                const uint64_t pa = *(const me_u64a4 *)(rp + 2 * r * rs), pb = *(const me_u64a4 *)(rp1 + 2 * r * rs);
                acc = svt_qsad(pa, s0[r], acc);
                acc = svt_qsad(pb, s1[r], acc);
            }
            pos[u] = (uint32_t)(ME_MUL(y, sw) + 4 * g);
            lo[u] = (uint32_t)acc; hi[u] = (uint32_t)(acc >> 32); /* positions pos, pos + 1 | pos + 2, pos + 3 as 16-bit sums */
        }
        _Pragma("unroll") for (int u = 0; u < NG; u++) {
            FP_KEYS(b8, lo[u], hi[u], pos[u]);
            /* 16x16: the quad's four blocks (sums stay below 2^16: no carry between the halves) */
            lo[u] = FP_DPP(lo[u], 0xB1); hi[u] = FP_DPP(hi[u], 0xB1); /* quad_perm:[1,0,3,2] */
            lo[u] = FP_DPP(lo[u], 0x4E); hi[u] = FP_DPP(hi[u], 0x4E); /* quad_perm:[2,3,0,1] */
            FP_KEYS(b16, lo[u], hi[u], pos[u]);
            /* 32x32: two quads still fit 16 bits; the other half of the row is added half by half into 32-bit sums */
            lo[u] = FP_DPP(lo[u], 0x124); hi[u] = FP_DPP(hi[u], 0x124); /* row_ror:4 */
            const uint32_t lo8 = (uint32_t)__builtin_amdgcn_mov_dpp((int)lo[u], 0x128, 0xf, 0xf, false); /* row_ror:8 */
            const uint32_t hi8 = (uint32_t)__builtin_amdgcn_mov_dpp((int)hi[u], 0x128, 0xf, 0xf, false);
            a0[u] = (lo[u] & 0xffffu) + (lo8 & 0xffffu); a1[u] = (lo[u] >> 16) + (lo8 >> 16);
            a2[u] = (hi[u] & 0xffffu) + (hi8 & 0xffffu); a3[u] = (hi[u] >> 16) + (hi8 >> 16);
            b32 = me_min3(b32, (a0[u] << 12) | pos[u], (a1[u] << 12) | (pos[u] + 1));
            b32 = me_min3(b32, (a2[u] << 12) | (pos[u] + 2), (a3[u] << 12) | (pos[u] + 3));
        }
        SVT_SCHED_FENCE(); /* the 32x32 keys above are done with a0..a3: the sums below run in place */
        /* 64x64: row 1 += row 0, row 3 += row 2 (row_bcast:15), then rows 2, 3 += row 1 (row_bcast:31): complete in lanes 48..63.
         * In place; the first DPP read comes two wait states behind the last write of its operand (s_nop: inline assembly is not
         * covered by the compiler's hazard recogniser), the second round reads what was written four instructions earlier. */
        _Pragma("unroll") for (int u = 0; u < NG; u++)
            __asm__("s_nop 1\n\t"
                    "v_add_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                    "v_add_u32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                    "v_add_u32_dpp %2, %2, %2 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                    "v_add_u32_dpp %3, %3, %3 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                    "v_add_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                    "v_add_u32_dpp %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                    "v_add_u32_dpp %2, %2, %2 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                    "v_add_u32_dpp %3, %3, %3 row_bcast:31 row_mask:0xc bank_mask:0xf"
                    : "+v"(a0[u]), "+v"(a1[u]), "+v"(a2[u]), "+v"(a3[u]));
        _Pragma("unroll") for (int u = 0; u < NG; u++) {
            b64 = me_min3(b64, (a0[u] << 12) | pos[u], (a1[u] << 12) | (pos[u] + 1));
            b64 = me_min3(b64, (a2[u] << 12) | (pos[u] + 2), (a3[u] << 12) | (pos[u] + 3));
        }
    }
#undef FP_KEYS
#undef FP_DPP
    uint64_t *key = c->st->key;
    if (b8 != 0xffffffffu) { /* this wave took at least one group */
        svt_lds_min_u64(&key[21 + z], ((uint64_t)((b8 >> 16) << 1) << 32) | (b8 & 0xffffu));
        if ((z & 3) == 0) svt_lds_min_u64(&key[5 + (z >> 2)], ((uint64_t)((b16 >> 16) << 1) << 32) | (b16 & 0xffffu));
        if ((z & 15) == 0) svt_lds_min_u64(&key[1 + (z >> 4)], ((uint64_t)((b32 >> 12) << 1) << 32) | (b32 & 0xfffu));
        if (z == 63) svt_lds_min_u64(&key[0], ((uint64_t)((b64 >> 12) << 1) << 32) | (b64 & 0xfffu));
    }
}
#endif

#ifndef SVT_HOST_EMU
/* The same phase for the LARGE areas whose width is a multiple of 16 (64 x 64 at the enc-mode <= 5 presets: 1024 groups of four positions
 * per list).  There the layout above is bound by the LDS, not by the vector unit: every group fetches its window again -- twelve 512-byte
 * LDS reads per group and wave, 8.5 cycles each with the two-way bank conflicts of the z-order: 104 K of the phase's 127 K cycles per list
 * (timing builds without the reads / without the QSADs: `profiles/r05_pmc_traffic.md`).  Two changes:
 *   - a lane walks a RUN of four groups (16 positions) along a search row: the six dwords of a window row serve all four (a group's two
 *     operand pairs overlap its neighbours'), three LDS reads instead of eight -- with the source block in registers 3 reads per group
 *     instead of 12;
 *   - a lane is a 16x16 PU (z-order) and one of FOUR runs (lane >> 4) and walks its four 8x8 blocks itself: the 16x16 sums are packed adds inside the lane, the 32x32 sums one packed and one
 *     32-bit quad step, the 64x64 sums two row rotations -- ~19 instructions per group beside its 8 QSADs instead of 48.
 *     (lane >> 4 picks one of four runs that lie UNDER each other, see the loop.)
 * The four runs' minima of a PU sit in four rows of the wave and meet at the end through a swizzle and two-way LDS minima (amortised over
 * the 16 iterations a wave runs per list; the small areas keep the layout above). */
SVT_DEV void me_fullpel_fused16_dev(const me_ctx_t *c, int tid, int sw, int sh) {
    const int rs = c->L.region_stride;
    const int lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pg = lane >> 4, b = lane & 15;
    const int bx = ((b & 1) | ((b >> 1) & 2)) * 16, by = (((b >> 1) & 1) | ((b >> 2) & 2)) * 16;
    /* The PUs at by and by + 32 sit in the same LDS banks whatever the row stride (32 rows are a multiple of 32 dwords), and a 32-lane pass
     * holds both: every window read was a two-way bank conflict (half of the phase's LDS-busy cycles, tools/me_phase_lds.sh).  The lower PUs
     * therefore walk their four rows one step ahead (row (r + 1) & 3 where the upper ones take row r: two rows = 70 dwords = 6 banks on, which
     * lands exactly in the banks the other half leaves free) -- a sum does not care about the order of its terms. */
    const int rot = by >> 5;
    uint32_t  sx[4][4], sy[4][4]; /* [8x8 block][step]: the two source dwords of row 2 ((step + rot) & 3) */
    uint32_t  roff[4];            /* byte offset of that row in the window */
    _Pragma("unroll") for (int r = 0; r < 4; r++) roff[r] = (uint32_t)ME_MUL(2 * ((r + rot) & 3), rs);
    _Pragma("unroll") for (int k = 0; k < 4; k++)
        _Pragma("unroll") for (int r = 0; r < 4; r++) {
            const uint2 v = *(const uint2 *)(c->src + ME_MUL(by + (k >> 1) * 8 + 2 * ((r + rot) & 3), ME_SB) + bx + (k & 1) * 8);
            sx[k][r] = v.x; sy[k][r] = v.y;
        }
    const uint32_t rbase = (uint32_t)(c->region - c->lds) + (uint32_t)(ME_MUL(ME_RGN_GY + by, rs) + ME_RGN_GX + bx);
    uint32_t mhi = 0xffff0000u;
    __asm__("" : "+v"(mhi));
    const int      rpr = sw >> 4, nrun = ME_MUL(rpr, sh);           /* runs of 16 positions per search row / in the area */
    const uint32_t inv = me_magics.v[sh];   /* runs are numbered down the columns: the four runs of an iteration lie under each other (see below) */
    uint32_t       b8[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu}, b16 = 0xffffffffu, b32 = 0xffffffffu, b64 = 0xffffffffu;
#define FQ_DPP(v, ctrl) ((v) + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), (ctrl), 0xf, 0xf, false))
#define FQ_KEYS(bb, lo, hi, pos) do { \
        bb = me_min3(bb, ((lo) << 16) | (pos), ((lo) & mhi) | ((pos) + 1)); \
        bb = me_min3(bb, ((hi) << 16) | ((pos) + 2), ((hi) & mhi) | ((pos) + 3)); } while (0)
    for (int q0 = 4 * w; q0 < nrun; q0 += 16) { /* the waves take four runs at a time, round-robin; a run past the end repeats the last one */
        const int      q = q0 + pg < nrun ? q0 + pg : nrun - 1;
        /* column-major: the lanes of the four runs then differ by whole region rows (39 dwords: every bank offset) instead of by 16 bytes,
         * which on top of the PUs' own 16-byte / 16-row spacing put most of a wave's reads into the same banks (bank-conflict cycles of the
         * phase halved; its time is the vector unit's either way) */
        const int      xr = inv ? (int)__umulhi((uint32_t)q, inv) : q, y = q - ME_MUL(xr, sh);
        const uint32_t pos0 = (uint32_t)(ME_MUL(y, sw) + 16 * xr);
        const uint8_t *rp = c->lds + (rbase + (uint32_t)(ME_MUL(y, rs) + 16 * xr));
        uint32_t       lo[4][4], hi[4][4]; /* [8x8 block][group of the run] */
        _Pragma("unroll") for (int k = 0; k < 4; k++) {
            const uint8_t *wp = rp + ME_MUL((k >> 1) * 8, rs) + (k & 1) * 8;
            uint64_t       acc[4] = {0, 0, 0, 0};
            _Pragma("unroll") for (int r = 0; r < 4; r++) {
                const uint32_t *wr = (const uint32_t *)(wp + roff[r]);
                const uint32_t  d0 = wr[0], d1 = wr[1], d2 = wr[2], d3 = wr[3], d4 = wr[4], d5 = wr[5];
                acc[0] = svt_qsad(((uint64_t)d1 << 32) | d0, sx[k][r], acc[0]); acc[0] = svt_qsad(((uint64_t)d2 << 32) | d1, sy[k][r], acc[0]);
                acc[1] = svt_qsad(((uint64_t)d2 << 32) | d1, sx[k][r], acc[1]); acc[1] = svt_qsad(((uint64_t)d3 << 32) | d2, sy[k][r], acc[1]);
                acc[2] = svt_qsad(((uint64_t)d3 << 32) | d2, sx[k][r], acc[2]); acc[2] = svt_qsad(((uint64_t)d4 << 32) | d3, sy[k][r], acc[2]);
                acc[3] = svt_qsad(((uint64_t)d4 << 32) | d3, sx[k][r], acc[3]); acc[3] = svt_qsad(((uint64_t)d5 << 32) | d4, sy[k][r], acc[3]);
            }
            _Pragma("unroll") for (int j = 0; j < 4; j++) {
                lo[k][j] = (uint32_t)acc[j]; hi[k][j] = (uint32_t)(acc[j] >> 32);
                FQ_KEYS(b8[k], lo[k][j], hi[k][j], pos0 + 4 * j);
            }
        }
        _Pragma("unroll") for (int j = 0; j < 4; j++) {
            const uint32_t pos = pos0 + 4 * j;
            /* 16x16: inside the lane (8 rows x 16 samples x 255 < 2^16: the packed halves do not carry) */
            uint32_t l16 = lo[0][j] + lo[1][j] + lo[2][j] + lo[3][j], h16 = hi[0][j] + hi[1][j] + hi[2][j] + hi[3][j];
            FQ_KEYS(b16, l16, h16, pos);
            /* 32x32: the quad.  Two PUs still fit 16 bits; the second step runs on 32-bit sums */
            l16 = FQ_DPP(l16, 0xB1); h16 = FQ_DPP(h16, 0xB1);                       /* quad_perm:[1,0,3,2] */
            uint32_t a0 = l16 & 0xffffu, a1 = l16 >> 16, a2 = h16 & 0xffffu, a3 = h16 >> 16;
            a0 = FQ_DPP(a0, 0x4E); a1 = FQ_DPP(a1, 0x4E); a2 = FQ_DPP(a2, 0x4E); a3 = FQ_DPP(a3, 0x4E); /* quad_perm:[2,3,0,1] */
            b32 = me_min3(b32, (a0 << 12) | pos, (a1 << 12) | (pos + 1));
            b32 = me_min3(b32, (a2 << 12) | (pos + 2), (a3 << 12) | (pos + 3));
            /* 64x64: the four quads of the run's row of 16 lanes */
            a0 = FQ_DPP(a0, 0x124); a1 = FQ_DPP(a1, 0x124); a2 = FQ_DPP(a2, 0x124); a3 = FQ_DPP(a3, 0x124); /* row_ror:4 */
            a0 = FQ_DPP(a0, 0x128); a1 = FQ_DPP(a1, 0x128); a2 = FQ_DPP(a2, 0x128); a3 = FQ_DPP(a3, 0x128); /* row_ror:8 */
            b64 = me_min3(b64, (a0 << 12) | pos, (a1 << 12) | (pos + 1));
            b64 = me_min3(b64, (a2 << 12) | (pos + 2), (a3 << 12) | (pos + 3));
        }
    }
#undef FQ_KEYS
#undef FQ_DPP
    /* the four runs' minima of a PU: rows pg and pg ^ 1 meet through a swizzle (lane ^ 16), the two halves of the wave in the LDS minimum */
#define FQ_X16(v) do { const uint32_t o_ = (uint32_t)__builtin_amdgcn_ds_swizzle((int)(v), 0x401F); v = o_ < v ? o_ : v; } while (0)
    _Pragma("unroll") for (int k = 0; k < 4; k++) FQ_X16(b8[k]);
    FQ_X16(b16); FQ_X16(b32); FQ_X16(b64);
#undef FQ_X16
    uint64_t *key = c->st->key;
    if ((pg & 1) == 0 && b16 != 0xffffffffu) { /* (a wave that took no run keeps nothing) */
        _Pragma("unroll") for (int k = 0; k < 4; k++) svt_lds_min_u64(&key[21 + 4 * b + k], ((uint64_t)((b8[k] >> 16) << 1) << 32) | (b8[k] & 0xffffu));
        svt_lds_min_u64(&key[5 + b], ((uint64_t)((b16 >> 16) << 1) << 32) | (b16 & 0xffffu));
        if ((b & 3) == 0) svt_lds_min_u64(&key[1 + (b >> 2)], ((uint64_t)((b32 >> 12) << 1) << 32) | (b32 & 0xfffu));
        if (b == 0) svt_lds_min_u64(&key[0], ((uint64_t)((b64 >> 12) << 1) << 32) | (b64 & 0xfffu));
    }
}
#endif

/* full-pel, search areas whose width is a multiple of 8 (no tail path) with at most 4096 positions: SADs, the nested sums and
 * the per-PU arg-min in ONE phase without the table.  Lane = 8x8 block in z-order, so a DPP quad is a 16x16 PU, a DPP row of
 * 16 lanes a 32x32 PU and the wave the 64x64 PU; the four waves take the groups of 4 positions round-robin.  A lane keeps one
 * running minimum per level as a 32-bit key -- (sad << 16) | position for 8x8 / 16x16 (sums < 2^16), (sad << 12) | position
 * for 32x32 / 64x64 -- and the waves meet in the same 64-bit LDS minimum as ph_fullpel_argmin: unsigned min = the
 * reference's first minimum in raster order. */
SVT_DEV void ph_fullpel_fused(const me_ctx_t *c, int tid, int sw, int sh, int unroll2) {
    (void)unroll2;
    const int rs = c->L.region_stride;
#ifdef SVT_HOST_EMU
    if (tid != 0) return;
    for (int y = 0; y < sh; y++)
        for (int x = 0; x < sw; x++) {
            uint32_t s8[64], s16[16], s32[4] = {0, 0, 0, 0}, s64 = 0;
            for (int z = 0; z < 64; z++) {
                const int bx = ((z & 1) | ((z >> 1) & 2) | ((z >> 2) & 4)) * 8, by = (((z >> 1) & 1) | ((z >> 2) & 2) | ((z >> 3) & 4)) * 8;
                uint32_t  a = 0;
                for (int r = 0; r < 8; r += 2)
                    for (int i = 0; i < 8; i++) {
                        const int d = (int)c->src[(by + r) * ME_SB + bx + i] - (int)c->region[(ME_RGN_GY + y + by + r) * rs + ME_RGN_GX + x + bx + i];
                        a += (uint32_t)(d < 0 ? -d : d);
                    }
                s8[z] = a;
            }
            for (int i = 0; i < 16; i++) s16[i] = (uint16_t)(s8[4 * i] + s8[4 * i + 1] + s8[4 * i + 2] + s8[4 * i + 3]);
            for (int i = 0; i < 16; i++) { s32[i >> 2] += s16[i]; s64 += s16[i]; }
            const uint32_t pos = (uint32_t)(y * sw + x);
            svt_lds_min_u64(&c->st->key[0], ((uint64_t)(2u * s64) << 32) | pos);
            for (int i = 0; i < 4; i++) svt_lds_min_u64(&c->st->key[1 + i], ((uint64_t)(2u * s32[i]) << 32) | pos);
            for (int i = 0; i < 16; i++) svt_lds_min_u64(&c->st->key[5 + i], ((uint64_t)(2u * s16[i]) << 32) | pos);
            for (int i = 0; i < 64; i++) svt_lds_min_u64(&c->st->key[21 + i], ((uint64_t)(2u * s8[i]) << 32) | pos);
        }
#else
    if (unroll2 && (sw & 15) == 0) me_fullpel_fused16_dev(c, tid, sw, sh);
    else if (unroll2) me_fullpel_fused_dev<2>(c, tid, sw, sh);
    else me_fullpel_fused_dev<2, true>(c, tid, sw, sh); /* (the phase's widths are multiples of 8: whole runs) */
#endif
}

/* full-pel: 16x16 sums of every position of the chunk.  In the 8-point path (x < w8) the reference keeps this sum
 * in uint16 (C_DEFAULT/EbComputeSAD_C.c:201,276), in the tail path in 32 bits. */
SVT_DEV void ph_fullpel_sum16(const me_ctx_t *c, int tid, uint32_t *U, int sw, int ny, int w8) {
    (void)c;
    int npos = sw * ny;
    for (int t = tid; t < npos * 16; t += SVT_NT) {
        int             pos = t >> 4, z = t & 15;
        const uint32_t *q   = U + pos * ME_PU_STRIDE + 21 + 2 * z;
        const uint32_t  q0 = q[0], q1 = q[1];
        uint32_t        u   = (q0 & 0xffffu) + (q0 >> 16) + (q1 & 0xffffu) + (q1 >> 16);
        if ((pos % sw) < w8) u = (uint16_t)u;
        U[pos * ME_PU_STRIDE + 5 + z] = u;
    }
}

/* full-pel: 32x32 sums (entries 1..4) and the 64x64 sum (entry 0) per position */
SVT_DEV void ph_fullpel_sum32(const me_ctx_t *c, int tid, uint32_t *U, int npos) {
    (void)c;
    for (int t = tid; t < npos * 5; t += SVT_NT) {
        int             pos = t / 5, j = t - 5 * pos;
        const uint32_t *q   = U + pos * ME_PU_STRIDE + 5;
        uint32_t        u   = 0;
        if (j < 4) u = q[4 * j] + q[4 * j + 1] + q[4 * j + 2] + q[4 * j + 3];
        else _Pragma("unroll") for (int i = 0; i < 16; i++) u += q[i];
        U[pos * ME_PU_STRIDE + (j < 4 ? 1 + j : 0)] = u;
    }
}

/* full-pel: per-PU arg-min.  Thread = (PU, one of 3 interleaved position slices); every PU reads the same table
 * layout, so the scan is branch-free and its loads are independent.  The slices meet in an LDS 64-bit min of
 * (2*sad << 32 | raster index): the unsigned min is exactly the reference's "first minimum in raster order"
 * (strict '<' while scanning positions in raster order). */
SVT_DEV void ph_fullpel_argmin(const me_ctx_t *c, int tid, const uint32_t *U, int sw, int y0, int ny) {
    const int npos = sw * ny;
    const int slice = tid / 85, pu = tid - 85 * slice;
    if (slice < 3) {
        uint32_t bsad = 0xffffffffu, bpos = 0;
        /* dword and bit field of this PU inside a table row */
        const uint32_t *q  = U + (pu < 21 ? pu : 21 + ((pu - 21) >> 1));
        const uint32_t  sh = pu < 21 ? 0u : (uint32_t)((pu - 21) & 1) * 16u, mk = pu < 21 ? 0xffffffffu : 0xffffu;
        for (int pos = slice; pos < npos; pos += 3) {
            uint32_t v = (q[pos * ME_PU_STRIDE] >> sh) & mk;
            if (v < bsad) { bsad = v; bpos = (uint32_t)pos; }
        }
        if (bsad != 0xffffffffu) svt_lds_min_u64(&c->st->key[pu], ((uint64_t)(2u * bsad) << 32) | (uint32_t)(y0 * sw + (int)bpos));
    }
}

SVT_DEV uint8_t me_clip8(int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }
SVT_DEV uint8_t me_tap4(int a, int b, int d, int e) { return me_clip8((-2 * a + 18 * b + 18 * d - 2 * e + 16) >> 5); }

/* 4-tap {-2,18,18,-2} (+16)>>5 with clipping on 4 packed samples: a,b,d,e hold the 4 taps of 4 neighbouring
 * outputs.  Even and odd bytes are processed as two 16-bit lanes of one register; a bias of 1024 (= 32 << 5) keeps
 * every lane non-negative so nothing borrows across lanes: floor((S + 1024) / 32) = floor(S / 32) + 32. */
SVT_DEV uint32_t me_tap4_half(uint32_t a, uint32_t b, uint32_t d, uint32_t e) {
#ifdef SVT_HOST_EMU
    const uint32_t s2 = (b + d) << 1;                          /* 18 x = 16 x + 2 x */
    uint32_t       v = (s2 << 3) + s2 + 0x04100410u - ((a + e) << 1); /* per lane: 18(b+d) + 16 + 1024 - 2(a+e) in [20, 10220] */
    v = (v >> 5) & 0x07ff07ffu;
    return svt_pk_clamp_sub32(v); /* per lane: min(max(v, 32), 287) - 32 */
#else
    /* on the packed 16-bit ALU, signed: two adds, two multiply-adds, an arithmetic shift; v_sat_pk_u8_i16 is the clip and
     * leaves the two samples in bytes 0 and 1 */
    typedef short   s16x2 __attribute__((ext_vector_type(2)));
    const s16x2     A = __builtin_bit_cast(s16x2, a), B = __builtin_bit_cast(s16x2, b), D = __builtin_bit_cast(s16x2, d), E = __builtin_bit_cast(s16x2, e);
    const s16x2     k5 = {5, 5};
    /* 18 (b + d) + 16, then - 2 (a + e) on top: two v_pk_mad_i16 (the compiler splits them into mul / shift / sub / add);
     * the value stays in [-1004, 9196] */
    uint32_t        m;
    __asm__("v_pk_mad_i16 %0, %1, 18, 16 op_sel_hi:[1,0,0]" : "=v"(m) : "v"(__builtin_bit_cast(uint32_t, B + D)));
    __asm__("v_pk_mad_i16 %0, %1, -2, %2 op_sel_hi:[1,0,1]" : "=v"(m) : "v"(__builtin_bit_cast(uint32_t, A + E)), "v"(m));
    s16x2           v = __builtin_bit_cast(s16x2, m) >> k5;
    uint32_t r;
    __asm__("v_sat_pk_u8_i16 %0, %1" : "=v"(r) : "v"(__builtin_bit_cast(uint32_t, v)));
    return r;
#endif
}
#ifndef SVT_HOST_EMU
/* even samples in bytes 0,1 of ev, odd ones in bytes 0,1 of od -> the 4 samples in order */
SVT_DEV uint32_t me_tap4_join(uint32_t ev, uint32_t od) { return __builtin_amdgcn_perm(od, ev, 0x05010400u); }
#endif
SVT_DEV uint32_t me_tap4_x4(uint32_t a, uint32_t b, uint32_t d, uint32_t e) {
#ifdef SVT_HOST_EMU
    const uint32_t M = 0x00ff00ffu;
    uint32_t ev = me_tap4_half(a & M, b & M, d & M, e & M);
    uint32_t od = me_tap4_half((a >> 8) & M, (b >> 8) & M, (d >> 8) & M, (e >> 8) & M);
    return ev | (od << 8);
#else
    /* even / odd bytes zero-extended into the two 16-bit lanes with one v_perm_b32 each (selector 0x0c = constant 0) */
    const uint32_t SE = 0x0c020c00u, SO = 0x0c030c01u;
    const uint32_t ev = me_tap4_half(__builtin_amdgcn_perm(0, a, SE), __builtin_amdgcn_perm(0, b, SE), __builtin_amdgcn_perm(0, d, SE), __builtin_amdgcn_perm(0, e, SE));
    const uint32_t od = me_tap4_half(__builtin_amdgcn_perm(0, a, SO), __builtin_amdgcn_perm(0, b, SO), __builtin_amdgcn_perm(0, d, SO), __builtin_amdgcn_perm(0, e, SO));
    return me_tap4_join(ev, od);
#endif
}

/* Half-pel planes (interpolate_search_region_avc, Codec/EbMotionEstimation.c:992-1070; C_DEFAULT/EbAvcStyleMcp_C.c:25-73), natural
 * coordinates with a guard of ME_PL_G samples: B (x + 1/2, y) = 4-tap filter along the region row, H (x, y + 1/2) = the same filter
 * down the region's columns, J (x + 1/2, y + 1/2) = the vertical filter over B, defined for y in [-1, H - 1].  Plane column px is
 * region column px + 2 (ME_RGN_GX - ME_PL_G), so the 7 region bytes a B dword needs sit in two aligned region dwords. */
/* ---- the three half-pel planes in ONE pass over column strips ----
 * A thread owns one dword column of the planes and a run of rows, and walks DOWN the region: region row R (two aligned dwords)
 * yields, in 16-bit lanes, the horizontal half-pel samples B(R - 1) and the samples the vertical filter needs from that row;
 * a window of the last four rows then gives H(R - 3) (vertical filter of the region) and J(R - 3) (vertical filter of B)
 * without re-reading anything: 2 LDS reads and 3 writes per output dword triple (a phase per plane pair needed 12 and 3, and
 * permuted every operand again for H and J). */
SVT_DEV uint32_t me_pair16(uint32_t hi, uint32_t lo, int k) { /* bytes k and k + 2 of the 8-byte pair, zero-extended into the two 16-bit lanes */
#ifdef SVT_HOST_EMU
    const uint64_t v = ((uint64_t)hi << 32) | lo;
    return (uint32_t)((v >> (8 * k)) & 0xff) | ((uint32_t)((v >> (8 * (k + 2))) & 0xff) << 16);
#else
    return __builtin_amdgcn_perm(hi, lo, 0x0c000c00u | (uint32_t)k | ((uint32_t)(k + 2) << 16));
#endif
}
SVT_DEV uint32_t me_half_lanes(uint32_t r) { /* result of me_tap4_half -> its two samples in the two 16-bit lanes */
#ifdef SVT_HOST_EMU
    return r;
#else
    return __builtin_amdgcn_perm(0, r, 0x0c010c00u);
#endif
}
SVT_DEV uint32_t me_half_join(uint32_t ev, uint32_t od) { /* even / odd results of me_tap4_half -> the four samples in order */
#ifdef SVT_HOST_EMU
    return (ev & 0xffu) | ((od & 0xffu) << 8) | (((ev >> 16) & 0xffu) << 16) | (((od >> 16) & 0xffu) << 24);
#else
    return me_tap4_join(ev, od);
#endif
}
SVT_DEV void ph_interp_strips(const me_ctx_t *c, int tid, int W, int H) {
    const int rs = c->L.region_stride, ps = c->L.plane_stride, pb = c->L.plane_bytes, pwd = (W + 2 * ME_PL_G + 3) >> 2, ph = H + 2 * ME_PL_G;
    const int nseg = me_udiv(SVT_NT, pwd), per = me_udiv(ph + nseg - 1, nseg);
    const int seg = me_udiv(tid, pwd), j = tid - seg * pwd;
    const int r0 = ME_MUL(seg, per), cnt = r0 + per < ph ? per : ph - r0; /* this strip: plane rows r0 .. r0 + cnt - 1 */
    if (seg >= nseg || cnt <= 0) return;
    /* Plane row py (natural row py - ME_PL_G) takes: B(py) from region row py + 1 (horizontal filter along it); H(py) and J(py)
     * from region rows py .. py + 3 -- H filters the rows themselves vertically, J the B rows derived from them (B(py - 1) ..
     * B(py + 2)).  Step i of the walk reads region row r0 + i (i = 0 .. cnt + 2), stores B(r0 + i - 1) and completes the window
     * of H / J(r0 + i - 3).  The window is a ring of four slots indexed by i & 3; the walk is unrolled by four so that every slot
     * is a named register (no moves) and the row offsets are immediates where the strides are. */
    uint32_t ve[4], vo[4], be[4], bo[4]; /* per window row: its samples for the vertical filter (even / odd), and the B row it yields */
    _Pragma("unroll") for (int k = 0; k < 4; k++) { ve[k] = vo[k] = be[k] = bo[k] = 0; }
    const uint8_t *rp = c->region + 4 * j + ME_MUL(r0, rs);      /* region row r0 + i0 */
    uint8_t       *wp = c->planes + 4 * j + ME_MUL(r0 - 3, ps);  /* plane row r0 + i0 - 3 of B (H, J: + pb, + 2 pb) */
    const int      jlo = 4 - r0, jhi = H + 5 - r0;               /* J exists for plane rows 1 .. H + 1: i in [jlo, jhi) */
    for (int i0 = 0; i0 < per + 3; i0 += 4) {                    /* same trip count in every lane; the stores carry the lane's bounds */
        _Pragma("unroll") for (int u = 0; u < 4; u++) {
            const int       i = i0 + u;
            if (i >= per + 3) break; /* (uniform: the walk is per + 3 steps long; unrolled by four it used to run up to three steps past its end) */
            const uint32_t *rw = (const uint32_t *)(rp + u * rs);
            const uint32_t  lo = rw[0], hi = rw[1];
            /* P(k) = bytes (k, k + 2) of the row's 8 bytes in 16-bit lanes: horizontal taps of the even outputs are P1..P4, of the
             * odd ones P2..P5; the vertical filter works on bytes 2..5 = P2 (even) and P3 (odd) */
            const uint32_t  p1 = me_pair16(hi, lo, 1), p2 = me_pair16(hi, lo, 2), p3 = me_pair16(hi, lo, 3), p4 = me_pair16(hi, lo, 4), p5 = me_pair16(hi, lo, 5);
            const uint32_t  he = me_tap4_half(p1, p2, p3, p4), ho = me_tap4_half(p2, p3, p4, p5);
            if (i >= 1 && i <= cnt) *(uint32_t *)(wp + (u + 2) * ps) = me_half_join(he, ho);
            ve[u] = p2; vo[u] = p3; be[u] = me_half_lanes(he); bo[u] = me_half_lanes(ho);
            if (i >= 3 && i < cnt + 3) {
                const int o = (u + 1) & 3, a = (u + 2) & 3, b = (u + 3) & 3; /* oldest .. newest = o, a, b, u */
                *(uint32_t *)(wp + u * ps + pb) = me_half_join(me_tap4_half(ve[o], ve[a], ve[b], ve[u]), me_tap4_half(vo[o], vo[a], vo[b], vo[u]));
                if (i >= jlo && i < jhi)
                    *(uint32_t *)(wp + u * ps + 2 * pb) = me_half_join(me_tap4_half(be[o], be[a], be[b], be[u]), me_tap4_half(bo[o], bo[a], bo[b], bo[u]));
            }
        }
        rp += 4 * rs; wp += 4 * ps;
    }
}

enum { ME_PF = 0, ME_PB = 1, ME_PH = 2, ME_PJ = 3 };
/* byte pointer (LDS) of plane `id` at natural position (x, y) relative to the region's top-left */
SVT_DEV int me_plane_stride(const me_ctx_t *c, int id) { return id == ME_PF ? c->L.region_stride : c->L.plane_stride; }
SVT_DEV const uint8_t *me_plane_at(const me_ctx_t *c, int id, int x, int y) {
    if (id == ME_PF) return c->region + ME_MUL(ME_RGN_GY + y, c->L.region_stride) + ME_RGN_GX + x;
    return c->planes + ME_MUL(id - 1, c->L.plane_bytes) + ME_MUL(y + ME_PL_G, c->L.plane_stride) + x + ME_PL_G;
}

/* SAD of a w x rows block: src rows at stride ss (LDS, dword aligned) vs candidate at any byte alignment (stride csa,
 * a multiple of 4), optionally averaged with a second candidate plane (b != 0, stride csb).  Each candidate row is fetched as
 * w/4 + 1 aligned dwords and shifted into place with v_alignbyte.  ssd_out != 0: also the sum of squared differences
 * (eb_vp9_spatial_full_distortion_kernel, C_DEFAULT/EbPictureOperators_C.c:337-356; averaging form
 * Codec/EbMotionEstimation.c:1708-1725). */
SVT_DEV uint32_t me_block_sad_rows(const uint8_t *src, int ss, const uint8_t *a, const uint8_t *b, int csa, int csb, int w, int r0, int r1, uint32_t *ssd_out) {
    uint32_t        sad = 0, ssd = 0;
    const uint32_t  sha = (uint32_t)((uintptr_t)a & 3), shb = b ? (uint32_t)((uintptr_t)b & 3) : 0;
    const uint8_t  *a0 = a - sha, *b0 = b ? b - shb : a0;
    const int       n = w >> 2;
    for (int r = r0; r < r1; r++) {
        const uint32_t *s  = (const uint32_t *)(src + ME_MUL(r, ss));
        const uint32_t *pa = (const uint32_t *)(a0 + ME_MUL(r, csa)), *pb = (const uint32_t *)(b0 + ME_MUL(r, csb));
        uint32_t        la = pa[0], lb = b ? pb[0] : 0;
        for (int i = 0; i < n; i++) {
            uint32_t ha = pa[i + 1];
            uint32_t va = svt_alignbyte(ha, la, sha);
            la = ha;
            if (b) {
                uint32_t hb = pb[i + 1];
                uint32_t vb = svt_alignbyte(hb, lb, shb);
                lb = hb;
                va = svt_avg4(va, vb); /* per-byte (a + b + 1) >> 1 */
            }
            sad = svt_sad4(va, s[i], sad);
            if (ssd_out) ssd = svt_ssd4(va, s[i], ssd);
        }
    }
    if (ssd_out) *ssd_out = ssd;
    return sad;
}

#ifdef SVT_HOST_EMU /* reference form of the packed candidate tables (checked by me_tables_selfcheck) */
/* candidate tables ---------------------------------------------------------------------------------- */
/* half-pel candidates L,R,T,B,TL,TR,BR,BL relative to the integer position (pu_half_pel_refinement,
 * Codec/EbMotionEstimation.c:1076-1559), natural coordinates */
__attribute__((unused)) static
    const int8_t me_hcand[8][3] = {{ME_PB, -1, 0}, {ME_PB, 0, 0}, {ME_PH, 0, -1}, {ME_PH, 0, 0},
                                   {ME_PJ, -1, -1}, {ME_PJ, 0, -1}, {ME_PJ, 0, 0}, {ME_PJ, -1, 0}};
__attribute__((unused)) static
    const int8_t me_hdmv[8][2] = {{-2, 0}, {2, 0}, {0, -2}, {0, 2}, {-2, -2}, {2, -2}, {2, 2}, {-2, 2}};
/* quarter-pel pairs (set_quarter_pel_refinement_inputs_on_the_fly, :2290-2465), natural coordinates
 * relative to P = (mv + 2) >> 2; [method][position L,R,T,B,TL,TR,BR,BL][plane1,dx1,dy1,plane2,dx2,dy2] */
__attribute__((unused)) static
    const int8_t me_qtab[4][8][6] = {
        {{ME_PB, -1, 0, ME_PF, 0, 0}, {ME_PF, 0, 0, ME_PB, 0, 0}, {ME_PH, 0, -1, ME_PF, 0, 0}, {ME_PF, 0, 0, ME_PH, 0, 0},
         {ME_PB, -1, 0, ME_PH, 0, -1}, {ME_PH, 0, -1, ME_PB, 0, 0}, {ME_PH, 0, 0, ME_PB, 0, 0}, {ME_PB, -1, 0, ME_PH, 0, 0}},
        {{ME_PF, -1, 0, ME_PB, -1, 0}, {ME_PB, -1, 0, ME_PF, 0, 0}, {ME_PJ, -1, -1, ME_PB, -1, 0}, {ME_PB, -1, 0, ME_PJ, -1, 0},
         {ME_PH, -1, -1, ME_PB, -1, 0}, {ME_PB, -1, 0, ME_PH, 0, -1}, {ME_PB, -1, 0, ME_PH, 0, 0}, {ME_PH, -1, 0, ME_PB, -1, 0}},
        {{ME_PJ, -1, -1, ME_PH, 0, -1}, {ME_PH, 0, -1, ME_PJ, 0, -1}, {ME_PF, 0, -1, ME_PH, 0, -1}, {ME_PH, 0, -1, ME_PF, 0, 0},
         {ME_PB, -1, -1, ME_PH, 0, -1}, {ME_PH, 0, -1, ME_PB, 0, -1}, {ME_PH, 0, -1, ME_PB, 0, 0}, {ME_PB, -1, 0, ME_PH, 0, -1}},
        {{ME_PH, -1, -1, ME_PJ, -1, -1}, {ME_PJ, -1, -1, ME_PH, 0, -1}, {ME_PB, -1, -1, ME_PJ, -1, -1}, {ME_PJ, -1, -1, ME_PB, -1, 0},
         {ME_PH, -1, -1, ME_PB, -1, -1}, {ME_PB, -1, -1, ME_PH, 0, -1}, {ME_PB, -1, 0, ME_PH, 0, -1}, {ME_PH, -1, -1, ME_PB, -1, 0}}};
__attribute__((unused)) static
    const int8_t me_qdmv[8][2] = {{-1, 0}, {1, 0}, {0, -1}, {0, 1}, {-1, -1}, {1, -1}, {1, 1}, {-1, 1}};
/* bi-pred quarter-pel compensation pairs (quarter_pel_compensation, :3358-3453), by frac_pos */
__attribute__((unused)) static
    const int8_t me_btab[16][6] = {
        {ME_PF, 0, 0, -1, 0, 0}, {ME_PF, 0, 0, ME_PB, 0, 0}, {ME_PB, 0, 0, -1, 0, 0}, {ME_PB, 0, 0, ME_PF, 1, 0},
        {ME_PF, 0, 0, ME_PH, 0, 0}, {ME_PB, 0, 0, ME_PH, 0, 0}, {ME_PB, 0, 0, ME_PJ, 0, 0}, {ME_PB, 0, 0, ME_PH, 1, 0},
        {ME_PH, 0, 0, -1, 0, 0}, {ME_PH, 0, 0, ME_PJ, 0, 0}, {ME_PJ, 0, 0, -1, 0, 0}, {ME_PJ, 0, 0, ME_PH, 1, 0},
        {ME_PH, 0, 0, ME_PF, 0, 1}, {ME_PH, 0, 0, ME_PB, 0, 1}, {ME_PJ, 0, 0, ME_PB, 0, 1}, {ME_PH, 1, 0, ME_PB, 0, 1}};

/* returns 0 when every packed / arithmetic table decodes to the reference tables above */
static inline int me_tables_selfcheck(void) {
    for (int i = 0; i < 64; i++) if (me_z8(i) != me_tab8x8[i] || me_inv8x8[me_z8(i)] != i) return 1;
    for (int i = 0; i < 16; i++) if (me_z4(i) != me_tab32x32[i] || me_inv32x32[me_z4(i)] != i) return 2;
    for (int i = 0; i < 8; i++) {
        int pl, dx, dy, sx, sy;
        me_hcand_get(i, &pl, &dx, &dy);
        if (pl != me_hcand[i][0] || dx != me_hcand[i][1] || dy != me_hcand[i][2]) return 3;
        me_dmv_get(i, &sx, &sy);
        if (2 * sx != me_hdmv[i][0] || 2 * sy != me_hdmv[i][1] || sx != me_qdmv[i][0] || sy != me_qdmv[i][1]) return 4;
    }
    for (int m = 0; m < 4; m++)
        for (int i = 0; i < 8; i++) {
            uint32_t       v = me_qtab_get(m, i);
            const int8_t *e = me_qtab[m][i];
            if ((int)(v & 3) != e[0] || -(int)((v >> 2) & 1) != e[1] || -(int)((v >> 3) & 1) != e[2]) return 5;
            if ((int)((v >> 4) & 3) != e[3] || -(int)((v >> 6) & 1) != e[4] || -(int)((v >> 7) & 1) != e[5]) return 6;
        }
    for (int f = 0; f < 16; f++) {
        int            hb;
        uint32_t       v = me_btab_get(f, &hb);
        const int8_t *e = me_btab[f];
        if ((int)(v & 3) != e[0] || (int)((v >> 2) & 1) != e[1] || (int)((v >> 3) & 1) != e[2]) return 7;
        if (hb != (e[3] >= 0)) return 8;
        if (hb && ((int)((v >> 4) & 3) != e[3] || (int)((v >> 6) & 1) != e[4] || (int)((v >> 7) & 1) != e[5])) return 9;
    }
    return 0;
}
#endif

/* direction codes, Codec/EbMotionEstimation.c:34-41 */
enum { ME_D_TL = 0, ME_D_T = 1, ME_D_TR = 2, ME_D_R = 3, ME_D_BR = 4, ME_D_B = 5, ME_D_BL = 6, ME_D_L = 7 };

/* which PUs are refined for the current list (half_pel_search_sb :1565-1702 gating) */
SVT_DEV int me_pu_refined(const me_ctx_t *c, int pu, int en32, int en16, int en8) {
    if (pu == 0) return c->p->fractional_search64x64;
    if (pu < 5) return en32;
    if (pu < 21) return en16 && c->p->cu16x16_mode == 0;
    return en8 && c->p->cu8x8_mode != 1;
}

/* lanes cooperating on one candidate block (row-interleaved) */
#define ME_SUB_LANES 8

/* the refined PUs of the current list as a dense index space: k in [0, me_active_count) -> raster pu */
SVT_DEV int me_active_count(const me_ctx_t *c, int en32, int en16, int en8, int *n64, int *n32, int *n16) {
    *n64 = c->p->fractional_search64x64 ? 1 : 0;
    *n32 = en32 ? 4 : 0;
    *n16 = (en16 && c->p->cu16x16_mode == 0) ? 16 : 0;
    return *n64 + *n32 + *n16 + ((en8 && c->p->cu8x8_mode != 1) ? 64 : 0);
}
SVT_DEV int me_active_pu(int k, int n64, int n32, int n16) {
    if (k < n64) return 0;
    k -= n64;
    if (k < n32) return 1 + k;
    k -= n32;
    if (k < n16) return 5 + k;
    return 21 + k - n16;
}

/* one record per refined PU so that the candidate tasks start from two LDS reads instead of re-deriving the PU from
 * its dense index (range tests, z-order interleave) under divergent branches */
SVT_DEV void ph_subpel_prep(const me_ctx_t *c, int tid, int en32, int en16, int en8) {
    int       n64, n32, n16;
    const int nact = me_active_count(c, en32, en16, en8, &n64, &n32, &n16);
    for (int k = tid; k < nact; k += SVT_NT) {
        const int pu = me_active_pu(k, n64, n32, n16);
        int       px, py, w;
        me_pu_geom(pu, &px, &py, &w);
        c->st->spu[k] = (uint32_t)pu | ((uint32_t)me_pu_nidx(pu) << 7) | ((uint32_t)(px >> 3) << 14) | ((uint32_t)(py >> 3) << 17) |
                        ((uint32_t)(w == 8 ? 0 : w == 16 ? 1 : w == 32 ? 2 : 3) << 20);
    }
}
#define ME_SPU_PU(i) ((int)((i) & 127))
#define ME_SPU_N(i) ((int)(((i) >> 7) & 127))
#define ME_SPU_PX(i) ((int)(((i) >> 14) & 7) << 3)
#define ME_SPU_PY(i) ((int)(((i) >> 17) & 7) << 3)
#define ME_SPU_W(i) (8 << (((i) >> 20) & 3))

/* Sub-pel work split: a candidate block of a 64x64 PU is shared by 16 lanes, of a 32x32 PU by 4 lanes, a 16x16 or 8x8
 * candidate is one lane's job (8 or 4 rows of 16 or 8 samples) -- at the BASELINE settings that is exactly 256 tasks of
 * equal size per half-pel pass.  The dense PU index k runs 64x64, 32x32, 16x16, 8x8 (me_active_pu), so the task ranges of
 * the three lane counts are contiguous.  t -> (k, candidate index, sub-lane, lanes per candidate); returns 0 past the end. */
/* lanes per candidate block: 16 for 64x64, 4 for 32x32, 1 below.  With the 21 PUs and 8 candidates of the M8 / M9 presets
 * that is 384 tasks = one and a half passes of the workgroup; 8 / 2 / 1 (exactly one pass of tasks twice as long) was
 * measured slower (ME 2.30 instead of 2.24 ms per mini-GOP): the longer serial row loops expose more LDS latency than the
 * half-empty second pass costs */
#define ME_HP_NL64 16
#define ME_HP_NL32 4
SVT_DEV int me_subpel_task(int t, int ncand, int n64, int n32, int nrest, int *k, int *ci, int *sl, int *nl) {
    const int T64 = n64 * ncand * ME_HP_NL64, T32 = n32 * ncand * ME_HP_NL32;
    int       q, base;
    if (t < T64) { *nl = ME_HP_NL64; *sl = t & (ME_HP_NL64 - 1); q = t / ME_HP_NL64; base = 0; }
    else if (t < T64 + T32) { const int u = t - T64; *nl = ME_HP_NL32; *sl = u & (ME_HP_NL32 - 1); q = u / ME_HP_NL32; base = n64; }
    else { q = t - T64 - T32; *nl = 1; *sl = 0; base = n64 + n32; if (q >= nrest * ncand) return 0; }
    const int kk = ncand == 8 ? q >> 3 : ncand == 3 ? q / 3 : q / 9;
    *k = base + kk; *ci = q - kk * ncand;
    return 1;
}

/* The sub-pel candidate table: entry k = pu * 8 + candidate.  Entries of the PUs 0..20 are dwords; those of the 8x8 PUs (k >= 168,
 * refined only when cu8x8_mode != 1) are halfwords -- an 8x8 SAD is at most 64 x 255 -- two to a dword at c->cand_hi: 1 KB instead of 2. */
SVT_DEV uint32_t me_cand_get(const me_ctx_t *c, int k) {
    if (k < 168) return c->cand[k];
    k -= 168;
    return (c->cand_hi[k >> 1] >> (16 * (k & 1))) & 0xffffu;
}
SVT_DEV uint32_t *me_cand_slot(const me_ctx_t *c, int k, int *shift) {
    if (k < 168) { *shift = 0; return &c->cand[k]; }
    k -= 168;
    *shift = 16 * (k & 1);
    return &c->cand_hi[k >> 1];
}
/* zero the table (and the candidate SSDs: keep_best = 1 leaves entry 8 of every PU, its best SSD so far) */
SVT_DEV void me_cand_zero(const me_ctx_t *c, int tid, int keep_best) {
    for (int t = tid; t < 85 * 9; t += SVT_NT) {
        if (t < (c->L.cand_dwords < 168 ? c->L.cand_dwords : 168)) c->cand[t] = 0;
        if (c->L.off_cand_hi >= 0 && t < 256) c->cand_hi[t] = 0;
        if (c->ssdc && !(keep_best && me_udiv(t, 9) * 9 + 8 == t)) c->ssdc[t] = 0;
    }
}

/* half-pel: distortion of every candidate accumulates in st->cand[pu*8+cand] (pre-zeroed).
 * SUB_SAD: rows 0,2,4.. only, doubled by the consumer; FULL_SAD: all rows.  SSD_SEARCH: 9 candidates per PU (8 = the
 * integer position, whose SSD seeds the comparison, :1107-1160), all rows, SAD in st->cand and SSD in c->ssdc. */
SVT_DEV void ph_halfpel(const me_ctx_t *c, int tid, int list, int sox, int soy, int en32, int en16, int en8) {
    const int sub_sad = c->p->fractional_search_method == SVT_SUB_SAD_SEARCH;
    const int ssd     = c->p->fractional_search_method == SVT_SSD_SEARCH;
    const int ncand   = ssd ? 9 : 8;
    int       n64, n32, n16;
    const int nact = me_active_count(c, en32, en16, en8, &n64, &n32, &n16);
    const int total = ncand * (n64 * ME_HP_NL64 + n32 * ME_HP_NL32 + (nact - n64 - n32));
    for (int t = tid; t < total; t += SVT_NT) {
        int k, cand, sl, nl;
        if (!me_subpel_task(t, ncand, n64, n32, nact - n64 - n32, &k, &cand, &sl, &nl)) break;
        const uint32_t info = c->st->spu[k];
        const int      pu = ME_SPU_PU(info), n = ME_SPU_N(info), px = ME_SPU_PX(info), py = ME_SPU_PY(info), w = ME_SPU_W(info);
        uint32_t mv = c->st->best_mv[list][n];
        int      xs = (int16_t)((me_mvx(mv) >> 2) - (int16_t)sox) + px;
        int      ys = (int16_t)((me_mvy(mv) >> 2) - (int16_t)soy) + py;
        int            hpl = ME_PF, hdx = 0, hdy = 0;
        if (cand < 8) me_hcand_get(cand, &hpl, &hdx, &hdy);
        const uint8_t *cp = me_plane_at(c, hpl, xs + hdx, ys + hdy);
        const uint8_t *sp = c->src + py * ME_SB + px;
        const int      rows = sub_sad ? (w >> 1) : w, step = sub_sad ? 2 : 1;
        const int      per = nl == ME_HP_NL64 ? rows / ME_HP_NL64 : nl == ME_HP_NL32 ? rows / ME_HP_NL32 : rows, r0 = sl * per;
        uint32_t e = 0;
        const int cs = me_plane_stride(c, hpl) * step;
        uint32_t d = me_block_sad_rows(sp, ME_SB * step, cp, 0, cs, cs, w, r0, r0 + per, ssd ? &e : 0);
        if (cand < 8) { int sh; uint32_t *slot = me_cand_slot(c, pu * 8 + cand, &sh); svt_group_add_var(slot, d << sh, nl); }
        if (ssd) svt_group_add_var(&c->ssdc[pu * 9 + cand], e, nl);
    }
}

/* half-pel decision per PU: sequential strict '<' updates in test order, then direction with the tie
 * order L,R,T,B,TL,TR,BL,BR (:1531-1556).  SSD_SEARCH compares SSDs and records the winner's SAD. */
SVT_DEV void ph_halfpel_decide(const me_ctx_t *c, int tid, int list, int en32, int en16, int en8) {
    const int sub_sad = c->p->fractional_search_method == SVT_SUB_SAD_SEARCH;
    const int ssd     = c->p->fractional_search_method == SVT_SSD_SEARCH;
    for (int pu = tid; pu < 85; pu += SVT_NT) {
        if (!me_pu_refined(c, pu, en32, en16, en8)) continue;
        int      n    = me_pu_nidx(pu);
        uint32_t best = c->st->best_sad[list][n], mv = c->st->best_mv[list][n];
        uint32_t bssd = ssd ? c->ssdc[pu * 9 + 8] : 0;
        int16_t  xm = me_mvx(mv), ym = me_mvy(mv);
        uint32_t d[8];
        for (int i = 0; i < 8; i++) {
            int sx, sy;
            me_dmv_get(i, &sx, &sy);
            if (ssd) {
                d[i] = c->ssdc[pu * 9 + i];
                if (d[i] < bssd) { bssd = d[i]; best = me_cand_get(c, pu * 8 + i); mv = me_pack_mv(xm + 2 * sx, ym + 2 * sy); }
            } else {
                d[i] = me_cand_get(c, pu * 8 + i);
                if (sub_sad) d[i] <<= 1;
                if (d[i] < best) { best = d[i]; mv = me_pack_mv(xm + 2 * sx, ym + 2 * sy); }
            }
        }
        uint32_t m = d[0];
        for (int i = 1; i < 8; i++) if (d[i] < m) m = d[i];
        uint8_t dir;
        if (m == d[0]) dir = ME_D_L;
        else if (m == d[1]) dir = ME_D_R;
        else if (m == d[2]) dir = ME_D_T;
        else if (m == d[3]) dir = ME_D_B;
        else if (m == d[4]) dir = ME_D_TL;
        else if (m == d[5]) dir = ME_D_TR;
        else if (m == d[7]) dir = ME_D_BL;
        else dir = ME_D_BR;
        c->st->best_sad[list][n] = best;
        c->st->best_mv[list][n]  = mv;
        c->st->dir[n]            = dir;
        if (ssd) c->ssdc[pu * 9 + 8] = bssd; /* (the thread that read the integer position's SSD there) */
    }
}

SVT_DEV int me_qvalid(int in_half, int dir, int pos) {
    /* pos: 0 L,1 R,2 T,3 B,4 TL,5 TR,6 BR,7 BL (:1761-1796) */
    int v_tl, v_t, v_tr, v_r, v_br, v_b, v_bl, v_l;
    if (in_half) {
        v_tl = dir == ME_D_R || dir == ME_D_BR || dir == ME_D_B;
        v_t  = dir == ME_D_BR || dir == ME_D_B || dir == ME_D_BL;
        v_tr = dir == ME_D_B || dir == ME_D_BL || dir == ME_D_L;
        v_r  = dir == ME_D_BL || dir == ME_D_L || dir == ME_D_TL;
        v_br = dir == ME_D_L || dir == ME_D_TL || dir == ME_D_T;
        v_b  = dir == ME_D_TL || dir == ME_D_T || dir == ME_D_TR;
        v_bl = dir == ME_D_T || dir == ME_D_TR || dir == ME_D_R;
        v_l  = dir == ME_D_TR || dir == ME_D_R || dir == ME_D_BR;
    } else {
        v_tl = dir == ME_D_L || dir == ME_D_TL || dir == ME_D_T;
        v_t  = dir == ME_D_TL || dir == ME_D_T || dir == ME_D_TR;
        v_tr = dir == ME_D_T || dir == ME_D_TR || dir == ME_D_R;
        v_r  = dir == ME_D_TR || dir == ME_D_R || dir == ME_D_BR;
        v_br = dir == ME_D_R || dir == ME_D_BR || dir == ME_D_B;
        v_b  = dir == ME_D_BR || dir == ME_D_B || dir == ME_D_BL;
        v_bl = dir == ME_D_B || dir == ME_D_BL || dir == ME_D_L;
        v_l  = dir == ME_D_BL || dir == ME_D_L || dir == ME_D_TL;
    }
    switch (pos) {
    case 0: return v_l; case 1: return v_r; case 2: return v_t; case 3: return v_b;
    case 4: return v_tl; case 5: return v_tr; case 6: return v_br; default: return v_bl;
    }
}

/* quarter-pel: task = (refined pu, j 0..2, sub-lane): the three positions around the half-pel direction.
 * The direction codes TL,T,TR,R,BR,B,BL,L run clockwise, and me_qvalid() accepts position X when
 * X is within one step of dir (integer best) or of the opposite of dir (half-pel best) (:1761-1796).
 * [quirk] the 64x64 PU is evaluated on its top-left 32x32 (:2525-2526). */
SVT_DEV void ph_quarterpel(const me_ctx_t *c, int tid, int list, int sox, int soy, int en32, int en16, int en8) {
    const int sub_sad = c->p->fractional_search_method == SVT_SUB_SAD_SEARCH;
    const int ssd     = c->p->fractional_search_method == SVT_SSD_SEARCH;
    int       n64, n32, n16;
    const int nact = me_active_count(c, en32, en16, en8, &n64, &n32, &n16);
    /* few candidates (3 per PU): 8 lanes share one candidate block so that the pass stays short */
    for (int t = tid; t < nact * 3 * ME_SUB_LANES; t += SVT_NT) {
        const int sl = t % ME_SUB_LANES, q = t / ME_SUB_LANES, k = q / 3, j = q - 3 * k, nl = ME_SUB_LANES;
        const uint32_t info = c->st->spu[k];
        const int      pu = ME_SPU_PU(info), n = ME_SPU_N(info), px = ME_SPU_PX(info), py = ME_SPU_PY(info);
        const int      w = pu == 0 ? 32 : ME_SPU_W(info);
        uint32_t mv = c->st->best_mv[list][n];
        int16_t  xm = me_mvx(mv), ym = me_mvy(mv);
        int      method = (ym & 2) + ((xm & 2) >> 1);
        int      dirx = ((method != 0 ? c->st->dir[n] ^ 4 : c->st->dir[n]) + j - 1) & 7;
        int      pos  = (int)(0x07361524u >> (4 * dirx)) & 7; /* direction code -> L,R,T,B,TL,TR,BR,BL index */
        int xs = (int16_t)(((xm + 2) >> 2) - (int16_t)sox) + px;
        int ys = (int16_t)(((ym + 2) >> 2) - (int16_t)soy) + py;
        const uint32_t e  = me_qtab_get(method, pos);
        const uint8_t *a  = me_plane_at(c, (int)(e & 3), xs - (int)((e >> 2) & 1), ys - (int)((e >> 3) & 1));
        const uint8_t *b  = me_plane_at(c, (int)((e >> 4) & 3), xs - (int)((e >> 6) & 1), ys - (int)((e >> 7) & 1));
        const uint8_t *sp = c->src + py * ME_SB + px;
        const int      rows = sub_sad ? (w >> 1) : w, step = sub_sad ? 2 : 1;
        const int      per = (rows + nl - 1) / nl, r0 = sl * per, r1 = r0 + per < rows ? r0 + per : rows;
        uint32_t sq = 0;
        uint32_t d = r0 < r1 ? me_block_sad_rows(sp, ME_SB * step, a, b, me_plane_stride(c, (int)(e & 3)) * step,
                                                 me_plane_stride(c, (int)((e >> 4) & 3)) * step, w, r0, r1, ssd ? &sq : 0) : 0;
        { int sh; uint32_t *slot = me_cand_slot(c, pu * 8 + pos, &sh); svt_group_add_u32(slot, d << sh, nl); }
        if (ssd) svt_group_add_u32(&c->ssdc[pu * 9 + pos], sq, nl);
    }
}

SVT_DEV void ph_quarterpel_decide(const me_ctx_t *c, int tid, int list, int en32, int en16, int en8) {
    const int sub_sad = c->p->fractional_search_method == SVT_SUB_SAD_SEARCH;
    const int ssd     = c->p->fractional_search_method == SVT_SSD_SEARCH;
    for (int pu = tid; pu < 85; pu += SVT_NT) {
        if (!me_pu_refined(c, pu, en32, en16, en8)) continue;
        int      n    = me_pu_nidx(pu);
        uint32_t best = c->st->best_sad[list][n], mv = c->st->best_mv[list][n];
        uint32_t bssd = ssd ? c->ssdc[pu * 9 + 8] : 0;
        int16_t  xm = me_mvx(mv), ym = me_mvy(mv);
        int      method = (ym & 2) + ((xm & 2) >> 1);
        int      dir = c->st->dir[n];
        for (int i = 0; i < 8; i++) {
            if (!me_qvalid(method != 0, dir, i)) continue;
            int sx, sy;
            me_dmv_get(i, &sx, &sy);
            if (ssd) {
                uint32_t e = c->ssdc[pu * 9 + i];
                if (e < bssd) { bssd = e; best = me_cand_get(c, pu * 8 + i); mv = me_pack_mv(xm + sx, ym + sy); }
            } else {
                uint32_t d = me_cand_get(c, pu * 8 + i);
                if (sub_sad) d <<= 1;
                if (d < best) { best = d; mv = me_pack_mv(xm + sx, ym + sy); }
            }
        }
        c->st->best_sad[list][n] = best;
        c->st->best_mv[list][n]  = mv;
        if (ssd) c->ssdc[pu * 9 + 8] = bssd;
    }
}

#ifndef SVT_HOST_EMU
/* ---- half- and quarter-pel refinement of the 32x32 and 16x16 PUs in ONE phase (SUB_SAD search, the M5+ presets) ----
 * The task lists above spend nine tenths of their instructions on finding out what a task is.  Here a lane owns 16 samples of
 * one (subsampled) row of one PU for the whole refinement: waves 0-1 the four 32x32 PUs (32 lanes each: 16 rows x 2 halves),
 * waves 2-3 the sixteen 16x16 PUs (8 lanes each: one row per lane).  The lane keeps its 4 source dwords and runs through the 8
 * half-pel candidates (planes and offsets are compile-time per candidate), the lanes of a PU are summed with DPP row shifts
 * (inclusive prefix: the PU's last lane holds the totals), that lane takes the reference's decisions (pu_half_pel_refinement
 * :1076-1559: strict '<' in test order = minimum of (distortion, test index); direction by the tie order L,R,T,B,TL,TR,BL,BR)
 * and publishes them through LDS -- LDS operations of one wave execute in order, so the PU's other lanes (same wave) read them
 * back without a barrier -- and the three quarter-pel candidates around that direction follow the same way
 * (pu_quarter_pel_refinement_on_the_fly :2471-2715).  No candidate table, no atomics, no barrier inside. */
SVT_DEV void me_pred_ptrs(const me_ctx_t *c, int list, int sox, int soy, int pu, int px, int py, const uint8_t **a, const uint8_t **b, int *sa, int *sb);
SVT_DEV uint32_t me_pred_fetch(const uint8_t *a, const uint8_t *b, int offa, int offb);
SVT_DEV uint32_t me_sad16(const uint8_t *p, const uint32_t s[4]) { /* 16 samples at any byte alignment in LDS against 4 source dwords */
    const uint32_t  sh = (uint32_t)((uintptr_t)p & 3);
    const uint32_t *q  = (const uint32_t *)(p - sh);
    const uint32_t  l0 = q[0], l1 = q[1], l2 = q[2], l3 = q[3], l4 = q[4];
    uint32_t        d = svt_sad4(svt_alignbyte(l1, l0, sh), s[0], 0);
    d = svt_sad4(svt_alignbyte(l2, l1, sh), s[1], d);
    d = svt_sad4(svt_alignbyte(l3, l2, sh), s[2], d);
    return svt_sad4(svt_alignbyte(l4, l3, sh), s[3], d);
}
SVT_DEV uint32_t me_sad16_avg(const uint8_t *pa, const uint8_t *pb, const uint32_t s[4]) { /* the same against the rounded average of two planes */
    const uint32_t  sa = (uint32_t)((uintptr_t)pa & 3), sb = (uint32_t)((uintptr_t)pb & 3);
    const uint32_t *qa = (const uint32_t *)(pa - sa), *qb = (const uint32_t *)(pb - sb);
    const uint32_t  a0 = qa[0], a1 = qa[1], a2 = qa[2], a3 = qa[3], a4 = qa[4], b0 = qb[0], b1 = qb[1], b2 = qb[2], b3 = qb[3], b4 = qb[4];
    uint32_t        d = svt_sad4(svt_avg4(svt_alignbyte(a1, a0, sa), svt_alignbyte(b1, b0, sb)), s[0], 0);
    d = svt_sad4(svt_avg4(svt_alignbyte(a2, a1, sa), svt_alignbyte(b2, b1, sb)), s[1], d);
    d = svt_sad4(svt_avg4(svt_alignbyte(a3, a2, sa), svt_alignbyte(b3, b2, sb)), s[2], d);
    return svt_sad4(svt_avg4(svt_alignbyte(a4, a3, sa), svt_alignbyte(b4, b3, sb)), s[3], d);
}
/* inclusive sums over the lanes of a PU (8 lanes, or 32 = two DPP rows): exact in the PU's last lane */
SVT_DEV uint32_t me_pu_lanes_sum(uint32_t v, int big) {
    v = SVT_DPP_ADD(v, 0x111); v = SVT_DPP_ADD(v, 0x112); v = SVT_DPP_ADD(v, 0x114);
    if (big) {
        v = SVT_DPP_ADD(v, 0x118);
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false); /* row_bcast:15 into rows 1 and 3 */
    }
    return v;
}
SVT_DEV void ph_subpel_fast(const me_ctx_t *c, int tid, int list, int sox, int soy, int en32, int en16, int bipred, uint32_t *pr) {
    me_state_t *st = c->st;
    const int   w = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63, big = w < 2;
    const int   refine = big ? en32 : en16; /* wave-uniform */
    if (!refine && !bipred) return;
    int pu, r, xo, px, py, last;
    if (big) { pu = 1 + 2 * w + (l >> 5); r = (l & 31) >> 1; xo = (l & 1) * 16; px = ((pu - 1) & 1) * 32; py = ((pu - 1) >> 1) * 32; last = (l & 31) == 31; }
    else { pu = 5 + 8 * (w - 2) + (l >> 3); r = l & 7; xo = 0; px = ((pu - 5) & 3) * 16; py = ((pu - 5) >> 2) * 16; last = (l & 7) == 7; }
    const int n = me_pu_nidx(pu), ps = c->L.plane_stride, pb = c->L.plane_bytes;
    uint32_t  s[4];
    {
        const uint32_t *sp = (const uint32_t *)(c->src + (py + 2 * r) * ME_SB + px + xo);
        s[0] = sp[0]; s[1] = sp[1]; s[2] = sp[2]; s[3] = sp[3];
    }
    uint32_t mv = st->best_mv[list][n], best = st->best_sad[list][n];
    int      xm = me_mvx(mv), ym = me_mvy(mv);
    /* ---- half-pel: 8 candidates ---- */
    if (refine) {
        const int      xs = (int16_t)((xm >> 2) - (int16_t)sox) + px + xo, ys = (int16_t)((ym >> 2) - (int16_t)soy) + py + 2 * r;
        const uint8_t *base = c->planes + ME_MUL(ys + ME_PL_G, ps) + xs + ME_PL_G; /* plane B at (xs, ys); H, J one / two planes further */
        uint32_t       d[8];
        _Pragma("unroll") for (int i = 0; i < 8; i++) {
            int hpl, hdx, hdy;
            me_hcand_get(i, &hpl, &hdx, &hdy);
            d[i] = me_sad16(base + (hpl - 1) * pb + hdy * ps + hdx, s);
        }
        /* a lane's sums stay below 2^12 and a group of 8 lanes below 2^15: two candidates per dword for the first three steps */
        uint32_t p4[4];
        _Pragma("unroll") for (int i = 0; i < 4; i++) {
            uint32_t v = d[i] | (d[i + 4] << 16);
            v = SVT_DPP_ADD(v, 0x111); v = SVT_DPP_ADD(v, 0x112); v = SVT_DPP_ADD(v, 0x114);
            p4[i] = v;
        }
        _Pragma("unroll") for (int i = 0; i < 4; i++) { d[i] = p4[i] & 0xffffu; d[i + 4] = p4[i] >> 16; }
        if (big) {
            _Pragma("unroll") for (int i = 0; i < 8; i++) {
                uint32_t v = SVT_DPP_ADD(d[i], 0x118);
                d[i] = v + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
            }
        }
        /* decisions (meaningful in the PU's last lane): distortions are doubled (rows 0, 2, 4, ...) */
        uint32_t km = 0xffffffffu, kr = 0xffffffffu;
        _Pragma("unroll") for (int i = 0; i < 8; i++) {
            const uint32_t dd = d[i] << 4; /* (2 d) << 3 */
            const uint32_t k1 = dd | (uint32_t)i, k2 = dd | (uint32_t)(i == 6 ? 7 : i == 7 ? 6 : i);
            km = k1 < km ? k1 : km; kr = k2 < kr ? k2 : kr;
        }
        if ((km >> 3) < best) {
            int sx, sy;
            me_dmv_get((int)(km & 7u), &sx, &sy);
            best = km >> 3; mv = me_pack_mv(xm + 2 * sx, ym + 2 * sy);
        }
        const uint32_t dir = (0x46205137u >> (4 * (kr & 7u))) & 7u; /* tie rank L,R,T,B,TL,TR,BL,BR -> direction code */
        if (last) { st->best_sad[list][n] = best; st->best_mv[list][n] = mv; st->dir[n] = (uint8_t)dir; }
    }
    __asm__ volatile("" ::: "memory"); /* the reads below must stay behind the stores above (other lanes' data) */
    /* ---- quarter-pel: the three positions around the half-pel direction ---- */
    if (refine) {
        mv = st->best_mv[list][n]; best = st->best_sad[list][n];
        const int dir = st->dir[n];
        xm = me_mvx(mv); ym = me_mvy(mv);
        const int method = (ym & 2) + ((xm & 2) >> 1);
        const int xs = (int16_t)(((xm + 2) >> 2) - (int16_t)sox) + px + xo, ys = (int16_t)(((ym + 2) >> 2) - (int16_t)soy) + py + 2 * r;
        uint32_t  q[3], pos[3];
        _Pragma("unroll") for (int j = 0; j < 3; j++) {
            const int dirx = ((method != 0 ? dir ^ 4 : dir) + j - 1) & 7;
            pos[j] = (0x07361524u >> (4 * dirx)) & 7u; /* direction code -> L,R,T,B,TL,TR,BR,BL index */
            const uint32_t e = me_qtab_get(method, (int)pos[j]);
            const uint8_t *a = me_plane_at(c, (int)(e & 3), xs - (int)((e >> 2) & 1), ys - (int)((e >> 3) & 1));
            const uint8_t *b = me_plane_at(c, (int)((e >> 4) & 3), xs - (int)((e >> 6) & 1), ys - (int)((e >> 7) & 1));
            q[j] = me_sad16_avg(a, b, s);
        }
        uint32_t v01 = q[0] | (q[1] << 16), v2 = q[2];
        v01 = SVT_DPP_ADD(v01, 0x111); v01 = SVT_DPP_ADD(v01, 0x112); v01 = SVT_DPP_ADD(v01, 0x114);
        v2 = SVT_DPP_ADD(v2, 0x111); v2 = SVT_DPP_ADD(v2, 0x112); v2 = SVT_DPP_ADD(v2, 0x114);
        q[0] = v01 & 0xffffu; q[1] = v01 >> 16; q[2] = v2;
        if (big) {
            _Pragma("unroll") for (int j = 0; j < 3; j++) {
                uint32_t v = SVT_DPP_ADD(q[j], 0x118);
                q[j] = v + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
            }
        }
        uint32_t km = 0xffffffffu;
        _Pragma("unroll") for (int j = 0; j < 3; j++) { const uint32_t k = (q[j] << 4) | pos[j]; km = k < km ? k : km; }
        if (last && (km >> 3) < best) {
            int sx, sy;
            me_dmv_get((int)(km & 7u), &sx, &sy);
            st->best_sad[list][n] = km >> 3; st->best_mv[list][n] = me_pack_mv(xm + sx, ym + sy);
        }
    }
    /* ---- the lane's 16 samples of the PU's prediction at its final motion vector (select_buffer :3310 / quarter_pel_compensation
     * :3358): kept in registers after list 0; after list 1 averaged with them and compared with the source -- the PU's
     * bi-prediction distortion (bi_pred_averging :3466-3560), summed over the PU's lanes, written by its last lane ---- */
    if (bipred) {
        __asm__ volatile("" ::: "memory");
        const uint8_t *a, *b;
        int            sa, sb;
        me_pred_ptrs(c, list, sox, soy, pu, px + xo, py + 2 * r, &a, &b, &sa, &sb);
        uint32_t v[4];
        {
            const uint32_t  sh = (uint32_t)((uintptr_t)a & 3);
            const uint32_t *q  = (const uint32_t *)(a - sh);
            const uint32_t  l0 = q[0], l1 = q[1], l2 = q[2], l3 = q[3], l4 = q[4];
            v[0] = svt_alignbyte(l1, l0, sh); v[1] = svt_alignbyte(l2, l1, sh); v[2] = svt_alignbyte(l3, l2, sh); v[3] = svt_alignbyte(l4, l3, sh);
        }
        if (b) {
            const uint32_t  sh = (uint32_t)((uintptr_t)b & 3);
            const uint32_t *q  = (const uint32_t *)(b - sh);
            const uint32_t  l0 = q[0], l1 = q[1], l2 = q[2], l3 = q[3], l4 = q[4];
            v[0] = svt_avg4(v[0], svt_alignbyte(l1, l0, sh)); v[1] = svt_avg4(v[1], svt_alignbyte(l2, l1, sh));
            v[2] = svt_avg4(v[2], svt_alignbyte(l3, l2, sh)); v[3] = svt_avg4(v[3], svt_alignbyte(l4, l3, sh));
        }
        if (list == 0) { pr[4] = v[0]; pr[5] = v[1]; pr[6] = v[2]; pr[7] = v[3]; }
        else {
            uint32_t d = svt_sad4(svt_avg4(pr[4], v[0]), s[0], 0);
            d = svt_sad4(svt_avg4(pr[5], v[1]), s[1], d);
            d = svt_sad4(svt_avg4(pr[6], v[2]), s[2], d);
            d = svt_sad4(svt_avg4(pr[7], v[3]), s[3], d);
            d = me_pu_lanes_sum(d, big);
            if (last) c->cand[pu] = d;
        }
        /* the 64x64 PU (never refined on this path: its vector is the full-pel one): every lane takes dword tid & 15 of the
         * subsampled rows 2 (tid >> 4) and 2 (tid >> 4) + 32; wave sums into cand[0], which the position-decode phase zeroed */
        const uint8_t *a0, *b0;
        int            sa0, sb0;
        me_pred_ptrs(c, list, sox, soy, 0, 0, 0, &a0, &b0, &sa0, &sb0);
        uint32_t d0 = 0;
        _Pragma("unroll") for (int k = 0; k < 2; k++) {
            const int      rr = 2 * (tid >> 4) + 32 * k, ii = tid & 15;
            const uint32_t vb = me_pred_fetch(a0, b0, ME_MUL(rr, sa0) + 4 * ii, ME_MUL(rr, sb0) + 4 * ii);
            if (list == 0) pr[k] = vb;
            else d0 = svt_sad4(svt_avg4(pr[k], vb), *(const uint32_t *)(c->src + rr * ME_SB + 4 * ii), d0);
        }
        if (list != 0) svt_wave_add_u32(&c->cand[0], d0, 1);
    }
}
#endif

/* Build the prediction block of the current list for every PU that takes part in bi-prediction
 * (select_buffer :3310 / quarter_pel_compensation :3358): task = (pu, row).  Output pred[pu_off + r*w + x].
 * Layout of pred blocks: pu 0 at 0 (64x64), 32x32 at 4096 + i*1024, 16x16 at 8192 + i*256, 8x8 at 12288 + i*64. */
SVT_DEV int me_pu_bipred(const me_ctx_t *c, int pu) {
    return (c->p->cu8x8_mode == 0 || pu < 21) && (c->p->cu16x16_mode == 0 || pu < 5);
}
/* prediction of `list` for a PU at its best mv: up to two source planes averaged (select_buffer :3310 /
 * quarter_pel_compensation :3358) */
SVT_DEV void me_pred_ptrs(const me_ctx_t *c, int list, int sox, int soy, int pu, int px, int py, const uint8_t **a, const uint8_t **b, int *sa, int *sb) {
    uint32_t mv = c->st->best_mv[list][me_pu_nidx(pu)];
    int16_t  mx = me_mvx(mv), my = me_mvy(mv);
    int      xi = (int16_t)(mx >> 2) - (int16_t)sox + px;
    int      yi = (int16_t)(my >> 2) - (int16_t)soy + py;
    int      frac = ((uint8_t)mx & 3) + (((uint8_t)my & 3) << 2);
    int            has_b;
    const uint32_t e = me_btab_get(frac, &has_b);
    *a = me_plane_at(c, (int)(e & 3), xi + (int)((e >> 2) & 1), yi + (int)((e >> 3) & 1));
    *b = has_b ? me_plane_at(c, (int)((e >> 4) & 3), xi + (int)((e >> 6) & 1), yi + (int)((e >> 7) & 1)) : 0;
    *sa = me_plane_stride(c, (int)(e & 3)); *sb = me_plane_stride(c, (int)((e >> 4) & 3));
}
SVT_DEV uint32_t me_pred_fetch(const uint8_t *a, const uint8_t *b, int offa, int offb) {
    uint32_t va = me_ld32u(a + offa);
    if (b) {
        uint32_t vb = me_ld32u(b + offb);
        va = svt_avg4(va, vb); /* (a + b + 1) >> 1 per byte */
    }
    return va;
}
SVT_DEV int me_bipred_levels(const me_ctx_t *c) { return c->p->cu16x16_mode != 0 ? 2 : c->p->cu8x8_mode != 0 ? 3 : 4; }

/* Bi-pred work split: level L (0 = 64x64 ... 3 = 8x8) has 4^L PUs of (1024 >> 2L) dwords; 256 >> 2L consecutive lanes
 * own one PU and each lane handles K dwords of it (K = 4, or 2 with SUB_SAD where only even rows count).  A lane meets
 * the same (level, k) dwords again when list 1 is searched: list 0's dword of (level, k) waits in the lane's own
 * registers pr[4 L + k] (at most 16; every index is a compile-time constant after unrolling).  The serial host
 * emulation keeps them in memory instead: ME_PR(j) = pred0[j * 256 + tid]. */
#ifdef SVT_HOST_EMU
#define ME_PR(j) pr[(j) * SVT_NT]
#else
#define ME_PR(j) pr[(j)]
#endif
SVT_DEV void ph_store_pred0(const me_ctx_t *c, int tid, int sox, int soy, uint32_t *pr, int lmax) {
    const int sub = c->p->fractional_search_method == SVT_SUB_SAD_SEARCH, K = sub ? 2 : 4, levels = me_bipred_levels(c) < lmax ? me_bipred_levels(c) : lmax;
    _Pragma("unroll") for (int L = 0; L < 4; L++) {
        if (L >= levels) break;
        const int sh = 8 - 2 * L, l = tid & ((1 << sh) - 1), pu = (int)((0x15050100u >> (8 * L)) & 0xff) + (tid >> sh);
        int       px, py, w;
        me_pu_geom(pu, &px, &py, &w);
        const uint8_t *a, *b;
        int sa, sb;
        me_pred_ptrs(c, 0, sox, soy, pu, px, py, &a, &b, &sa, &sb);
        _Pragma("unroll") for (int k = 0; k < 4; k++) {
            if (k < K) {
                int d = l + (k << sh), r = (d >> (4 - L)) << sub, i = d & ((16 >> L) - 1);
                ME_PR(4 * L + k) = me_pred_fetch(a, b, ME_MUL(r, sa) + 4 * i, ME_MUL(r, sb) + 4 * i);
            }
        }
        SVT_SCHED_FENCE();
    }
}
/* bi-pred distortion: avg-SAD of (list0 pred, list1 pred) vs source (bi_pred_averging :3466-3560) */
SVT_DEV void ph_bipred(const me_ctx_t *c, int tid, int sox, int soy, const uint32_t *pr, int lmax) {
    const int sub = c->p->fractional_search_method == SVT_SUB_SAD_SEARCH, K = sub ? 2 : 4, levels = me_bipred_levels(c) < lmax ? me_bipred_levels(c) : lmax;
    _Pragma("unroll") for (int L = 0; L < 4; L++) {
        if (L >= levels) break;
        const int sh = 8 - 2 * L, l = tid & ((1 << sh) - 1), pu = (int)((0x15050100u >> (8 * L)) & 0xff) + (tid >> sh);
        int       px, py, w;
        me_pu_geom(pu, &px, &py, &w);
        const uint8_t *a, *b;
        int sa, sb;
        me_pred_ptrs(c, 1, sox, soy, pu, px, py, &a, &b, &sa, &sb);
        uint32_t dsum = 0;
        _Pragma("unroll") for (int k = 0; k < 4; k++) {
            if (k < K) {
                int      d = l + (k << sh), r = (d >> (4 - L)) << sub, i = d & ((16 >> L) - 1);
                uint32_t s  = *(const uint32_t *)(c->src + (py + r) * ME_SB + px + 4 * i);
                uint32_t va = ME_PR(4 * L + k), vb = me_pred_fetch(a, b, ME_MUL(r, sa) + 4 * i, ME_MUL(r, sb) + 4 * i);
                uint32_t av = svt_avg4(va, vb);
                dsum = svt_sad4(av, s, dsum);
            }
        }
        svt_group_add_u32(&c->cand[pu], dsum, sh > 6 ? 64 : 1 << sh);
        SVT_SCHED_FENCE();
    }
}

/* candidate ordering + result record (Codec/EbMotionEstimation.c:5186-5293), one thread per PU */
SVT_DEV void ph_output(const me_ctx_t *c, int tid, svt_me_pu_result *out, uint32_t *out_words) {
    const int sub_sad = c->p->fractional_search_method == SVT_SUB_SAD_SEARCH;
    const int nlist   = c->p->num_ref_lists;
    (void)out;
    for (int pu = tid; pu < 85; pu += SVT_NT) {
        int      n = me_pu_nidx(pu);
        int      total = nlist;
        uint32_t l0 = c->st->best_sad[0][n], l1 = nlist == 2 ? c->st->best_sad[1][n] : 0, bi = 0;
        if (nlist == 2 && me_pu_bipred(c, pu)) {
            bi = c->cand[pu];
            if (sub_sad) bi <<= 1;
            total = 3;
        }
        uint32_t w[10];
        uint32_t mv0 = c->st->best_mv[0][n], mv1 = nlist == 2 ? c->st->best_mv[1][n] : 0;
        w[0] = mv0;
        w[1] = mv1;
        for (int i = 2; i < 10; i++) w[i] = 0;
        if (total == 3) {
            uint32_t v[3] = {l0, l1, bi};
            int      o[3];
            if (l0 <= l1 && l0 <= bi) { o[0] = 0; if (l1 <= bi) { o[1] = 1; o[2] = 2; } else { o[1] = 2; o[2] = 1; } }
            else if (l1 <= l0 && l1 <= bi) { o[0] = 1; if (l0 <= bi) { o[1] = 0; o[2] = 2; } else { o[1] = 2; o[2] = 0; } }
            else if (l0 <= l1) { o[0] = 2; o[1] = 0; o[2] = 1; }
            else { o[0] = 2; o[1] = 1; o[2] = 0; }
            for (int i = 0; i < 3; i++) { w[2 + 2 * i] = v[o[i]]; w[3 + 2 * i] = (uint32_t)o[i]; }
        } else if (total == 2) {
            if (l0 <= l1) { w[2] = l0; w[3] = 0; w[4] = l1; w[5] = 1; }
            else { w[2] = l1; w[3] = 1; w[4] = l0; w[5] = 0; }
        } else {
            w[2] = l0; w[3] = 0;
        }
        w[8] = (uint32_t)total;
        for (int i = 0; i < 10; i++) out_words[pu * 10 + i] = w[i];
    }
}

/* ------------------------------------------------------------------------------------------------ */
/* driver: uniform control flow; PHASE(x) runs x for every thread and ends with a workgroup barrier     */
/* ------------------------------------------------------------------------------------------------ */
/* ME_MARK(i): when profiling is enabled, thread 0 adds the shader cycles since the previous mark to prof[i] */
#if defined(SVT_HOST_EMU)
#define ME_MARK(i) ((void)0)
#define ME_SUBMARK_BEGIN() ((void)0)
#define ME_SUBMARK(i) ((void)0)
#define ME_STOP_AT(i) ((void)0)
#else
/* sub-phase marks (slots 14, 15): informational, not part of the per-phase total */
#define ME_SUBMARK_BEGIN() unsigned long long sub_t_ = c->prof ? __builtin_amdgcn_s_memtime() : 0
#define ME_SUBMARK(i) do { if (c->prof && tid == 0) { unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
        atomicAdd(&c->prof[(i)], now_ - sub_t_); sub_t_ = now_; } } while (0)
#ifdef ME_FINE_PROF
#define ME_STOP_AT(i) do { if (g_me_stop_after == (i)) return; } while (0)
#else
#define ME_STOP_AT(i) ((void)0)
#endif
#if defined(ME_ASM_MARKS)
/* static instruction counts (tools/me_static_counts.py): a comment in the assembly at every mark */
#define ME_MARK(i) __asm__ volatile("; @MARK %0" ::"n"(i))
#elif defined(ME_FINE_PROF)
/* instruction-count profiling builds: the kernel stops (all threads) at mark g_me_stop_after of the first list, so
 * that per-dispatch SQ counters of successive launches give cumulative instruction counts per phase */
__device__ int g_me_stop_after = -1;
#define ME_MARK(i) do { if (c->prof && tid == 0) { unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
        atomicAdd(&c->prof[(i)], now_ - mark_t_); mark_t_ = now_; } if (g_me_stop_after == (i)) return; } while (0)
#else
#define ME_MARK(i) do { if (c->prof && tid == 0) { unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
        atomicAdd(&c->prof[(i)], now_ - mark_t_); mark_t_ = now_; } } while (0)
#endif
#endif
#ifdef SVT_HOST_EMU
#define ME_PHASE(...) do { for (int tid = 0; tid < SVT_NT; tid++) { __VA_ARGS__; } } while (0)
#define ME_UNIFORM_WRITE(...) do { __VA_ARGS__; } while (0)
#define ME_UNI(x) ((int)(x))
#else
/* the thread index is re-read through an opaque move at every phase: whatever a phase derives from it (lane roles,
 * LDS addresses) is computed where it is used and dies with the phase, instead of being hoisted to the top of the
 * kernel and kept (or spilled) across all the others */
#define ME_PHASE(...) do { __asm__ volatile("" : "+v"(tid)); __VA_ARGS__; __syncthreads(); } while (0)
/* uniform state written to LDS by one thread, followed by a barrier */
#define ME_UNIFORM_WRITE(...) do { if (tid == 0) { __VA_ARGS__; } __syncthreads(); } while (0)
/* a value every lane holds identically (read from LDS): move it to a scalar register */
#define ME_UNI(x) __builtin_amdgcn_readfirstlane((int)(x))
#endif

/* t / d through inv = floor((2^32 - 1) / d) + 1 (exact while t * d < 2^32; d = 1 gives inv = 0 -> t).  Every thread derives
 * inv itself when it enters a window: the (slow) division runs in parallel instead of on the planning thread */
SVT_DEV uint32_t me_magic_of(int d) { return (uint32_t)(0xffffffffu / (uint32_t)d) + 1u; }
/* the same from the reciprocal table when d is small (no division) */
#ifdef SVT_HOST_EMU
static inline uint32_t me_magic_small(int d) { return me_magic_of(d); }
#else
/* (d is the same in every active lane -- the planning thread is alone: a scalar load through the constant cache instead of a vector
 * load with its ~1 us round trip on the critical path of the workgroup) */
SVT_DEV uint32_t me_magic_small(int d) { const int du = __builtin_amdgcn_readfirstlane(d); return du <= 256 ? me_magics.v[du] : me_magic_of(du); }
#endif
SVT_DEV int me_div_magic(int t, uint32_t inv) { return inv ? (int)(((uint64_t)(uint32_t)t * inv) >> 32) : t; }

/* copy the windows [e0, e1) of a batch: flattened (window, row, 16-byte unit) tasks -- one global load per unit (the last unit
 * of a row is shortened to whole dwords), two units in flight per thread before the LDS stores; ntask = total load tasks */
#define ME_HME_UNITS(nd) (((nd) + 3) >> 2)
SVT_DEV void ph_hme_load_multi(const me_ctx_t *c, int tid, const svt_plane *ref_lds, const me_hme_win *wn, int e0, int e1, int ntask) {
    const svt_plane  ref_u = me_plane_uni(ref_lds);
    const svt_plane *ref = &ref_u;
    for (int t0 = tid; t0 < ntask; t0 += 2 * SVT_NT) {
        me_u32x4 v[2];
        int      dst[2], k[2];
        _Pragma("unroll") for (int u = 0; u < 2; u++) {
            int T = t0 + u * SVT_NT;
            dst[u] = -1; k[u] = 0;
            if (T < ntask) {
                int e = e0;
                while (e + 1 < e1 && T >= wn[e + 1].tl) e++;
                const int t = T - wn[e].tl, nd = wn[e].nd, nu = ME_HME_UNITS(nd);
                const int row = me_div_magic(t, wn[e].inv_nu), i = t - row * nu;
                const uint8_t *gp = me_pix(ref, wn[e].gx + 16 * i, wn[e].gy + row);
                k[u] = nd - 4 * i < 4 ? nd - 4 * i : 4;
                if (k[u] == 4) v[u] = me_ld128u_g(gp);
                else {
                    v[u].x = me_ld32u_g(gp);
                    v[u].y = k[u] > 1 ? me_ld32u_g(gp + 4) : 0;
                    v[u].z = k[u] > 2 ? me_ld32u_g(gp + 8) : 0;
                    v[u].w = 0;
                }
                dst[u] = wn[e].off + row * wn[e].wstride + 16 * i;
            }
        }
        _Pragma("unroll") for (int u = 0; u < 2; u++)
            if (dst[u] >= 0) {
                uint32_t *d = (uint32_t *)(c->hme_scratch + dst[u]);
                d[0] = v[u].x;
                if (k[u] > 1) d[1] = v[u].y;
                if (k[u] > 2) d[2] = v[u].z;
                if (k[u] > 3) d[3] = v[u].w;
            }
    }
}

/* SADs of 4 consecutive search positions (window dwords wr..) against a bw x bh block; window row of block row j is
 * mul*j rows further down.  The packed u16 accumulators are flushed before they can overflow.  Two block rows are
 * processed per step with independent accumulators and all their LDS loads issued up front: the QSAD chain of one
 * row overlaps the other's (a dependent v_qsad_pk_u16_u8 costs ~26 cycles, an LDS round trip ~64+). */
SVT_DEV void me_qsad_row4(const uint32_t *wr, const uint32_t *br, uint64_t *acc) {
    const uint32_t w0 = wr[0], w1 = wr[1], w2 = wr[2], w3 = wr[3], w4 = wr[4];
    const uint32_t b0 = br[0], b1 = br[1], b2 = br[2], b3 = br[3];
    uint64_t       a = *acc;
    a = svt_qsad(((uint64_t)w1 << 32) | w0, b0, a);
    a = svt_qsad(((uint64_t)w2 << 32) | w1, b1, a);
    a = svt_qsad(((uint64_t)w3 << 32) | w2, b2, a);
    a = svt_qsad(((uint64_t)w4 << 32) | w3, b3, a);
    *acc = a;
}
SVT_DEV void me_qsad_block(const uint8_t *blk, int bstride, int nd, int bh, const uint8_t *win, int wstride, int mul, uint32_t a[4]) {
    const int flush = nd <= 4 ? 16 : nd <= 8 ? 8 : 4; /* rows whose sums (4*nd*255 each) still fit 16 bits */
    a[0] = a[1] = a[2] = a[3] = 0;
    for (int j0 = 0; j0 < bh; j0 += flush) {
        uint64_t  acc0 = 0, acc1 = 0; /* even / odd rows of the group: each holds at most flush/2 rows */
        const int j1  = j0 + flush < bh ? j0 + flush : bh;
        int       j   = j0;
        if (nd == 4) { /* 16-sample rows (1/16-resolution level): the whole row pair is loaded before the first QSAD */
            for (; j + 2 <= j1; j += 2) {
                const uint32_t *wa = (const uint32_t *)(win + mul * j * wstride), *wb = (const uint32_t *)(win + mul * (j + 1) * wstride);
                const uint32_t *ba = (const uint32_t *)(blk + j * bstride), *bb = (const uint32_t *)(blk + (j + 1) * bstride);
                const uint32_t  x0 = wa[0], x1 = wa[1], x2 = wa[2], x3 = wa[3], x4 = wa[4];
                const uint32_t  y0 = wb[0], y1 = wb[1], y2 = wb[2], y3 = wb[3], y4 = wb[4];
                const uint32_t  p0 = ba[0], p1 = ba[1], p2 = ba[2], p3 = ba[3], q0 = bb[0], q1 = bb[1], q2 = bb[2], q3 = bb[3];
                acc0 = svt_qsad(((uint64_t)x1 << 32) | x0, p0, acc0); acc1 = svt_qsad(((uint64_t)y1 << 32) | y0, q0, acc1);
                acc0 = svt_qsad(((uint64_t)x2 << 32) | x1, p1, acc0); acc1 = svt_qsad(((uint64_t)y2 << 32) | y1, q1, acc1);
                acc0 = svt_qsad(((uint64_t)x3 << 32) | x2, p2, acc0); acc1 = svt_qsad(((uint64_t)y3 << 32) | y2, q2, acc1);
                acc0 = svt_qsad(((uint64_t)x4 << 32) | x3, p3, acc0); acc1 = svt_qsad(((uint64_t)y4 << 32) | y3, q3, acc1);
            }
        }
        for (; j < j1; j++) {
            const uint32_t *wr = (const uint32_t *)(win + mul * j * wstride);
            const uint32_t *br = (const uint32_t *)(blk + j * bstride);
            uint64_t        acc = (j & 1) ? acc1 : acc0;
            int             i = 0;
            for (; i + 4 <= nd; i += 4) me_qsad_row4(wr + i, br + i, &acc);
            if (i < nd) {
                uint32_t lo = wr[i];
                for (; i < nd; i++) {
                    uint32_t hi = wr[i + 1];
                    acc         = svt_qsad(((uint64_t)hi << 32) | lo, br[i], acc);
                    lo          = hi;
                }
            }
            if (j & 1) acc1 = acc; else acc0 = acc;
        }
        a[0] += (uint32_t)(acc0 & 0xffff) + (uint32_t)(acc1 & 0xffff);
        a[1] += (uint32_t)((acc0 >> 16) & 0xffff) + (uint32_t)((acc1 >> 16) & 0xffff);
        a[2] += (uint32_t)((acc0 >> 32) & 0xffff) + (uint32_t)((acc1 >> 32) & 0xffff);
        a[3] += (uint32_t)(acc0 >> 48) + (uint32_t)(acc1 >> 48);
    }
}

/* 32-bit form of the HME key for the 1/16-resolution level: (sad << 16) | (y << 8) | x -- the SAD of a 16 x 8 block is below
 * 2^15 and the search positions of a region stay below 256 either way; ordered exactly like the 64-bit key it stands for */
SVT_DEV uint64_t me_hme_key64(uint32_t k) { return ((uint64_t)(k >> 16) << 32) | (((k >> 8) & 0xffu) << 16) | (k & 0xffu); }
#ifdef SVT_HOST_EMU
static inline void svt_wave_min_key32(uint64_t *p, uint32_t k) { if (k != 0xffffffffu && me_hme_key64(k) < *p) *p = me_hme_key64(k); }
#else
/* min over the wave (all lanes must call; ~0 = nothing), then ONE 64-bit LDS atomic by lane 0 */
SVT_DEV void svt_wave_min_key32(uint64_t *p, uint32_t k) {
#define SVT_DPP_MIN32(ctrl) do { const uint32_t o_ = (uint32_t)__builtin_amdgcn_update_dpp((int)k, (int)k, (ctrl), 0xf, 0xf, false); k = o_ < k ? o_ : k; } while (0)
    SVT_DPP_MIN32(0x111); SVT_DPP_MIN32(0x112); SVT_DPP_MIN32(0x114); SVT_DPP_MIN32(0x118);
#undef SVT_DPP_MIN32
    uint32_t m = (uint32_t)__builtin_amdgcn_readlane((int)k, 15);
    _Pragma("unroll") for (int l = 31; l < 64; l += 16) { const uint32_t w = (uint32_t)__builtin_amdgcn_readlane((int)k, l); m = w < m ? w : m; }
    if ((threadIdx.x & 63) == 0 && m != 0xffffffffu)
        __hip_atomic_fetch_min((unsigned long long *)p, (unsigned long long)me_hme_key64(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
#endif

/* the 1/16-resolution level with a whole SB (16 x 8 block, rows of 4 dwords; window rows two apart): straight-line task --
 * each QSAD operand pair is read as such (the overlapping pairs cost LDS reads, not register moves), one packed add joins the
 * even / odd row accumulators (8 rows x 16 samples x 255 < 2^16) */
SVT_DEV void me_qsad_16x8(const uint8_t *blk, const uint8_t *win, int wstride, uint32_t *lo_out, uint32_t *hi_out) {
    uint64_t acc0 = 0, acc1 = 0;
    _Pragma("unroll") for (int j = 0; j < 8; j++) {
        const uint32_t *w = (const uint32_t *)(win + 2 * j * wstride), *b = (const uint32_t *)(blk + 16 * j);
        uint64_t        a = (j & 1) ? acc1 : acc0;
        _Pragma("unroll") for (int i = 0; i < 4; i++) a = svt_qsad(((uint64_t)w[i + 1] << 32) | w[i], b[i], a);
        if (j & 1) acc1 = a; else acc0 = a;
    }
    *lo_out = (uint32_t)acc0 + (uint32_t)acc1;                 /* positions 0, 1 as 16-bit sums: no carry between the halves */
    *hi_out = (uint32_t)(acc0 >> 32) + (uint32_t)(acc1 >> 32); /* positions 2, 3 */
}

/* exhaustive search of the windows [e0, e1) of a batch in one phase; keys[slot] = min over
 * (sad << 32 | y << 16 | x inside the region): ordered like the raster index, no division to take it apart */
#ifndef SVT_HOST_EMU
/* The windows of a batch (at most four here: one per region, or the bands of one), held in scalar registers: a task finds its window
 * by three comparisons instead of walking the list in LDS -- every task used to start with up to five dependent LDS round trips
 * (~120 cycles each with one wave per SIMD) before its first sample was fetched. */
typedef struct me_hme_sel { int ts[4], sw[4], slot[4], y0[4], ws[4], off[4]; uint32_t inv[4]; } me_hme_sel;
SVT_DEV void me_hme_sel_load(me_hme_sel *S, const me_hme_win *wn, int e0, int e1) {
    _Pragma("unroll") for (int k = 0; k < 4; k++) {
        const me_hme_win *w = &wn[e0 + k < e1 ? e0 + k : e1 - 1];
        S->ts[k] = e0 + k < e1 ? ME_UNI(w->ts) : 0x7fffffff;
        S->sw[k] = ME_UNI(w->sw); S->slot[k] = ME_UNI(w->slot); S->y0[k] = ME_UNI(w->y0); S->ws[k] = ME_UNI(w->wstride);
        S->off[k] = ME_UNI(w->off); S->inv[k] = (uint32_t)ME_UNI(w->inv_ng);
    }
}
#define ME_HME_SEL(S, T, f) ((T) >= (S).ts[3] ? (S).f[3] : (T) >= (S).ts[2] ? (S).f[2] : (T) >= (S).ts[1] ? (S).f[1] : (S).f[0])
#endif
SVT_DEV void ph_hme_search_multi(const me_ctx_t *c, int tid, const uint8_t *blk, int bstride, int bw, int bh, const me_hme_win *wn,
                                 int e0, int e1, int ntask, uint64_t *keys, int slot_mask) {
    const int qs = (bw & 3) == 0; /* QSAD path: task = 4 positions */
    uint64_t  best[4] = {~0ull, ~0ull, ~0ull, ~0ull}; /* per key slot */
    int       cur = -1;
    uint32_t  inv = 0;
#ifdef ME_FINE_PROF
    unsigned long long ft_ = __builtin_amdgcn_s_memtime();
#define FP(i) do { if (c->prof && tid == 0) { unsigned long long n_ = __builtin_amdgcn_s_memtime(); atomicAdd(&c->prof[i], n_ - ft_); ft_ = n_; } } while (0)
#else
#define FP(i) ((void)0)
#endif
    if (bw == 16 && bh == 8 && bstride == 16 && c->L.hme_tw0 <= 256 && c->L.hme_th0 <= 256 && c->L.hme_w0[0] <= 256 && c->L.hme_w0[1] <= 256 &&
        c->L.hme_h0[0] <= 256 && c->L.hme_h0[1] <= 256) { /* positions inside a region fit 8 bits each */
        uint32_t b32[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
#ifndef SVT_HOST_EMU
        me_hme_sel S;
        const bool sel = e1 - e0 <= 4;
        if (sel) me_hme_sel_load(&S, wn, e0, e1);
#endif
        for (int T = tid; T < ntask; T += SVT_NT) {
            int      t, sw, slot, w_y0, w_ws, w_off;
            uint32_t w_inv;
#ifndef SVT_HOST_EMU
            if (sel) {
                t = T - ME_HME_SEL(S, T, ts); sw = ME_HME_SEL(S, T, sw); slot = ME_HME_SEL(S, T, slot); w_y0 = ME_HME_SEL(S, T, y0);
                w_ws = ME_HME_SEL(S, T, ws); w_off = ME_HME_SEL(S, T, off); w_inv = ME_HME_SEL(S, T, inv);
            } else
#endif
            {
                int e = e0;
                while (e + 1 < e1 && T >= wn[e + 1].ts) e++;
                t = T - wn[e].ts; sw = wn[e].sw; slot = wn[e].slot; w_y0 = wn[e].y0; w_ws = wn[e].wstride; w_off = (int)wn[e].off; w_inv = wn[e].inv_ng;
            }
            const int ng = (sw + 3) >> 2, y = me_div_magic(t, w_inv), g = t - ME_MUL(y, ng);
            uint32_t  lo, hi;
            me_qsad_16x8(blk, c->hme_scratch + w_off + ME_MUL(y, w_ws) + 4 * g, w_ws, &lo, &hi);
            const uint32_t pos = ((uint32_t)(w_y0 + y) << 8) | (uint32_t)(4 * g);
            uint32_t       k0 = (lo << 16) | pos, k1 = (lo & 0xffff0000u) | (pos + 1), k2 = (hi << 16) | (pos + 2), k3 = (hi & 0xffff0000u) | (pos + 3);
            if (4 * g + 3 >= sw) { /* last group of a width that is not a multiple of 4 */
                if (4 * g + 1 >= sw) k1 = 0xffffffffu;
                if (4 * g + 2 >= sw) k2 = 0xffffffffu;
                k3 = 0xffffffffu;
            }
            k0 = k0 < k1 ? k0 : k1; k2 = k2 < k3 ? k2 : k3; k0 = k0 < k2 ? k0 : k2;
            _Pragma("unroll") for (int q = 0; q < 4; q++) if (q == slot && k0 < b32[q]) b32[q] = k0;
        }
        _Pragma("unroll") for (int q = 0; q < 4; q++) if ((slot_mask >> q) & 1) svt_wave_min_key32(&keys[q], b32[q]);
        return;
    }
#ifndef SVT_HOST_EMU
    if (qs && (bw == 32 || bw == 64) && (bh == 16 || bh == 32)) {
        /* The quarter- and full-resolution levels of a whole SB: few tasks (a region is 4 x 2 .. 16 x 16 positions) of many samples each
         * (32 x 16 / 64 x 32 rows) -- one lane per task left three waves idle while a handful of lanes walked 128 / 512 QSADs each (the
         * 1080p presets: 8 lanes busy for ~20 K cycles).  Here bw / 4 = 8 or 16 neighbouring lanes share a task: a lane owns one dword
         * column of the block over all its rows (4 positions x <= 32 rows x 4 samples x 255 stay below 2^16 per 16-bit sum), the
         * columns meet in DPP row shifts (the group's last lane holds the four sums), and that lane keeps the task's key. */
        const int lsh = bw == 64 ? 4 : 3, nl = 1 << lsh, sub = tid & (nl - 1), tpp = SVT_NT >> lsh;
        const uint8_t *bcol = blk + 4 * sub;
        me_hme_sel     S;
        const bool     sel = e1 - e0 <= 4;
        if (sel) me_hme_sel_load(&S, wn, e0, e1);
        for (int T0 = 0; T0 < ntask; T0 += tpp) {
            const int  T = T0 + (tid >> lsh);
            const bool act = T < ntask;
            int        a0 = 0, a1 = 0, a2 = 0, a3 = 0, slot = 0, sw = 0, y = 0, g = 0, y0 = 0;
            if (act) {
                int      t, ws, w_off;
                uint32_t w_inv;
                if (sel) {
                    t = T - ME_HME_SEL(S, T, ts); sw = ME_HME_SEL(S, T, sw); slot = ME_HME_SEL(S, T, slot); y0 = ME_HME_SEL(S, T, y0);
                    ws = ME_HME_SEL(S, T, ws); w_off = ME_HME_SEL(S, T, off); w_inv = ME_HME_SEL(S, T, inv);
                } else {
                    int e = e0;
                    while (e + 1 < e1 && T >= wn[e + 1].ts) e++;
                    t = T - wn[e].ts; ws = wn[e].wstride; w_off = (int)wn[e].off; w_inv = wn[e].inv_ng;
                    sw = wn[e].sw; slot = wn[e].slot; y0 = wn[e].y0;
                }
                const int ng = (sw + 3) >> 2;
                y = me_div_magic(t, w_inv); g = t - ME_MUL(y, ng);
                const uint8_t *wp = c->hme_scratch + w_off + ME_MUL(y, ws) + 4 * g + 4 * sub;
                uint64_t       acc = 0, acc_b = 0; /* two chains; eight rows' operands are fetched before their QSADs (bh is 16 or 32) */
                const int      ws2 = 2 * ws;
                for (int j = 0; j < bh; j += 8) {
                    uint64_t pr[8];
                    uint32_t bd[8];
                    _Pragma("unroll") for (int u = 0; u < 8; u++) {
                        pr[u] = *(const me_u64a4 *)(wp + ME_MUL(j + u, ws2));
                        bd[u] = *(const uint32_t *)(bcol + ME_MUL(j + u, bstride));
                    }
                    _Pragma("unroll") for (int u = 0; u < 8; u += 2) { acc = svt_qsad(pr[u], bd[u], acc); acc_b = svt_qsad(pr[u + 1], bd[u + 1], acc_b); }
                }
                acc += acc_b; /* (16-bit sums of 32 rows x 4 samples: no carry between the fields) */
                a0 = (int)(acc & 0xffffu); a1 = (int)((acc >> 16) & 0xffffu); a2 = (int)((acc >> 32) & 0xffffu); a3 = (int)(acc >> 48);
            }
            /* every lane takes part (lanes without a task add 0); row_shr:n with bound_ctrl: lanes shifted in from outside the row read 0 */
#define HW_SHR(v, n) v += __builtin_amdgcn_update_dpp(0, v, 0x110 + (n), 0xf, 0xf, true)
            HW_SHR(a0, 1); HW_SHR(a1, 1); HW_SHR(a2, 1); HW_SHR(a3, 1);
            HW_SHR(a0, 2); HW_SHR(a1, 2); HW_SHR(a2, 2); HW_SHR(a3, 2);
            HW_SHR(a0, 4); HW_SHR(a1, 4); HW_SHR(a2, 4); HW_SHR(a3, 4);
            if (lsh == 4) { HW_SHR(a0, 8); HW_SHR(a1, 8); HW_SHR(a2, 8); HW_SHR(a3, 8); }
#undef HW_SHR
            if (act && sub == nl - 1) {
                const uint32_t av[4] = {(uint32_t)a0, (uint32_t)a1, (uint32_t)a2, (uint32_t)a3};
                uint64_t       kb = ~0ull;
                _Pragma("unroll") for (int o = 0; o < 4; o++) {
                    const int x = 4 * g + o;
                    if (x < sw) {
                        const uint64_t k = ((uint64_t)av[o] << 32) | ((uint32_t)(y0 + y) << 16) | (uint32_t)x;
                        if (k < kb) kb = k;
                    }
                }
                _Pragma("unroll") for (int q = 0; q < 4; q++) if (q == slot && kb < best[q]) best[q] = kb;
            }
        }
        _Pragma("unroll") for (int q = 0; q < 4; q++) if ((slot_mask >> q) & 1) svt_wave_min_u64(&keys[q], best[q]);
        return;
    }
#endif
    for (int T = tid; T < ntask; T += SVT_NT) {
        int e = e0;
        while (e + 1 < e1 && T >= wn[e + 1].ts) e++;
        const int      t = T - wn[e].ts, ws = wn[e].wstride, sw = wn[e].sw, slot = wn[e].slot, y0 = wn[e].y0;
        if (e != cur) { cur = e; inv = wn[e].inv_ng; }
        const uint8_t *win = c->hme_scratch + wn[e].off;
        uint64_t       kb = ~0ull;
        FP(16);
        if (qs) {
            const int ng = (sw + 3) >> 2;
            const int y = me_div_magic(t, inv), g = t - y * ng;
            uint32_t  a[4];
            me_qsad_block(blk, bstride, bw >> 2, bh, win + y * ws + 4 * g, ws, 2, a);
            FP(17);
            _Pragma("unroll") for (int o = 0; o < 4; o++) {
                int x = 4 * g + o;
                if (x < sw) {
                    uint64_t k = ((uint64_t)a[o] << 32) | ((uint32_t)(y0 + y) << 16) | (uint32_t)x;
                    if (k < kb) kb = k;
                }
            }
        } else {
            const int y = me_div_magic(t, inv), x = t - y * sw;
            uint32_t  sd = 0;
            for (int j = 0; j < bh; j++)
                for (int i = 0; i < bw; i++) {
                    int p0 = blk[j * bstride + i], p1 = win[(y + 2 * j) * ws + x + i];
                    sd += (uint32_t)(p0 > p1 ? p0 - p1 : p1 - p0);
                }
            kb = ((uint64_t)sd << 32) | ((uint32_t)(y0 + y) << 16) | (uint32_t)x;
        }
        _Pragma("unroll") for (int q = 0; q < 4; q++) if (q == slot && kb < best[q]) best[q] = kb;
        FP(18);
    }
    /* every lane takes part in the wave reductions of the slots this batch touches (lanes without work contribute ~0) */
    _Pragma("unroll") for (int q = 0; q < 4; q++) if ((slot_mask >> q) & 1) svt_wave_min_u64(&keys[q], best[q]);
    FP(19);
#undef FP
}

typedef struct me_hme_geom {
    const svt_plane *ref;
    const uint8_t   *blk; /* LDS */
    int              bstride, bw, bh, ox, oy, pad_w, pad_h, ref_w, ref_h;
} me_hme_geom;

/* a value the planning thread (alone in its wave) reads from LDS: on the device it goes to a scalar register, so that the
 * arithmetic built on it -- placement, clipping, window sizes: everything a level's plan computes -- runs on the scalar unit
 * instead of as a chain of dependent vector instructions of one lane (the workgroup waits for this thread: with the one or two
 * waves per SIMD the 64x64-area configurations leave, its latency is not hidden by anything) */
#ifdef SVT_HOST_EMU
#define ME_PLAN_RD(x) (x)
#else
#define ME_PLAN_RD(x) __builtin_amdgcn_readfirstlane((int)(x))
#endif
SVT_DEV int16_t me_hme_round_w(int16_t w) { return (int16_t)((w < 8) ? 8 : (w & 7) ? w + (w - ((w >> 3) << 3)) : w); }

/* geometry of an HME level for one reference list (hme_level0/1/2 of Codec/EbMotionEstimation.c) */
SVT_DEV void me_hme_geom_of(const me_ctx_t *c, int list, int lvl, me_hme_geom *g) {
    if (lvl == 0) {
        g->ref = &c->st->refd[2]; g->blk = c->st->sixteenth_sb; g->bstride = 16; g->bw = c->sb_w >> 2; g->bh = (c->sb_h >> 2) >> 1;
        g->ox = (int16_t)(c->sb_x >> 2); g->oy = (int16_t)(c->sb_y >> 2);
    } else if (lvl == 1) {
        g->ref = &c->st->refd[1]; g->blk = c->quarter_sb; g->bstride = 64; g->bw = c->sb_w >> 1; g->bh = (c->sb_h >> 1) >> 1;
        g->ox = (int16_t)(c->sb_x >> 1); g->oy = (int16_t)(c->sb_y >> 1);
    } else {
        g->ref = &c->st->refd[0]; g->blk = c->src; g->bstride = 2 * ME_SB; g->bw = c->sb_w; g->bh = c->sb_h >> 1;
        g->ox = (int16_t)c->sb_x; g->oy = (int16_t)c->sb_y;
    }
    g->pad_w = lvl == 2 ? ME_SB - 1 : ME_PLAN_RD(g->ref->origin_x) - 1;
    g->pad_h = lvl == 2 ? ME_SB - 1 : ME_PLAN_RD(g->ref->origin_y) - 1;
    g->ref_w = ME_PLAN_RD(g->ref->width); g->ref_h = ME_PLAN_RD(g->ref->height);
}

/* Plan one HME level (run by ONE thread): place the search areas of the level's regions (slot = rh*2 + rw), clip them,
 * and cut them into the work list of (region, band of search rows) windows.  Consecutive windows that fit the scratch
 * together form a batch = one global-load phase + one search phase; a region too tall for the scratch is split into
 * bands -- the 64-bit key carries the raster index inside the region, so the minimum over all bands is exactly the
 * reference's first minimum in raster order.  [quirk] region-row counter semantics: see me_sb_run. */
SVT_DEV void me_hme_plan_level(const me_ctx_t *c, int list, int lvl, int16_t xsc, int16_t ysc, int first) {
    const svt_me_params *p  = c->p;
    me_state_t          *st = c->st;
    const int            NW = p->number_hme_search_region_in_width, NH = p->number_hme_search_region_in_height;
    me_hme_geom          g;
    me_hme_geom_of(c, list, lvl, &g);
    const int single = lvl == 0 && p->single_hme_quadrant && !p->enable_hme_level_1_flag && !p->enable_hme_level_2_flag;
    const int span   = 2 * (g.bh - 1);
    int       ne = 0, nb = 0, bytes = 0, tl = 0, ts = 0;
#ifndef SVT_HOST_EMU
    if (single) {
        /* one region, one level (the 4K presets M8+): only slot 0 of level 0 is ever read back (me_hme_finish_level / me_hme_select
         * skip the other slots) -- the same steps as the general code below for k = 0, without the loops around them */
        st->hme_x[0][0] = (int16_t)(xsc >> 2); st->hme_y[0][0] = (int16_t)(ysc >> 2);
        st->hme_rh = 0;
        st->hme_bstart[0] = 0;
        { uint32_t *kw_ = (uint32_t *)&st->hme_keys[0]; kw_[0] = ~0u; kw_[1] = ~0u; }
        int16_t w = c->L.hme_tw0, h = c->L.hme_th0;
        int16_t ox = (int16_t)(-(int16_t)(w >> 1) + (int16_t)(xsc >> 2)), oy = (int16_t)(-(int16_t)(h >> 1) + (int16_t)(ysc >> 2));
        me_clip_area(g.ox, &ox, &w, g.pad_w, g.ref_w);
        me_clip_area(g.oy, &oy, &h, g.pad_h, g.ref_h);
        if ((w & 15) != 0) w = (int16_t)((w >> 4) << 4);
        st->hme_cox[0] = ox; st->hme_coy[0] = oy;
        const int ok = w > 0 && h > 0;
        st->hme_cw[0] = ok ? w : 0; st->hme_ch[0] = ok ? h : 0;
        if (ok) {
            const int wbytes = w + g.bw + 3;
            int       ws     = ((wbytes + 3) & ~3) + 4;
            if (((ws >> 2) & 1) == 0) ws += 4;
            const int ng = (g.bw & 3) == 0 ? (w + 3) >> 2 : w;
            for (int y = 0; y < h && ne < ME_HME_MAX_WIN;) {
                int nr = h - y;
                if (ws * (nr + span) > c->hme_scratch_bytes - bytes) nr = (c->hme_scratch_bytes - bytes) / ws - span;
                if (nr < 1 && bytes > 0) { st->hme_bstart[++nb] = ne; bytes = 0; tl = 0; ts = 0; continue; }
                if (nr < 1) break;
                me_hme_win *wn = &st->hme_win[ne++];
                wn->off = bytes; wn->wstride = ws; wn->nd = (wbytes + 3) >> 2; wn->rows = nr + span; wn->sw = w; wn->sh = nr;
                wn->gx = g.ox + ox; wn->gy = g.oy + oy + y; wn->slot = 0; wn->y0 = y; wn->tl = tl; wn->ts = ts;
                wn->inv_nu = me_magic_small(ME_HME_UNITS(wn->nd)); wn->inv_ng = me_magic_small(ng);
                tl += ME_HME_UNITS(wn->nd) * wn->rows; ts += ng * nr;
                bytes += ws * (nr + span); y += nr;
            }
        }
        if (ne > st->hme_bstart[nb]) st->hme_bstart[++nb] = ne;
        st->hme_nbatch = nb;
        return;
    }
#endif
    if (first && ME_PLAN_RD(st->hme_rh) < NH) { /* [quirk] centres are only initialised while the reference's row counter is below NH */
        for (int k = 0; k < 4; k++)
            if ((k & 1) < NW && (k >> 1) < NH && (k >> 1) >= ME_PLAN_RD(st->hme_rh)) {
                st->hme_x[0][k] = (int16_t)(xsc >> 2); st->hme_y[0][k] = (int16_t)(ysc >> 2);
                st->hme_x[1][k] = (int16_t)(xsc >> 1); st->hme_y[1][k] = (int16_t)(ysc >> 1);
                st->hme_x[2][k] = xsc; st->hme_y[2][k] = ysc;
            }
        st->hme_rh = NH;
    }
    st->hme_rh = single ? 0 : NH;
    st->hme_bstart[0] = 0;
    for (int k = 0; k < 4; k++) {
        const int rw = k & 1, rh = k >> 1;
        {   /* written as two dwords: as a 64-bit constant the compiler hoists the pair out of the SB's whole life and spills it */
            uint32_t *kw_ = (uint32_t *)&st->hme_keys[k];
            kw_[0] = ~0u; kw_[1] = ~0u;
        }
        st->hme_cw[k] = 0; st->hme_ch[k] = 0; st->hme_cox[k] = 0; st->hme_coy[k] = 0;
        if (single ? k != 0 : (rw >= NW || rh >= NH)) continue;
        int16_t w, h, ox, oy;
        if (lvl == 0) {
            /* c->L.hme_* = (area * multiplier) / 100 of hme_level0 / single_hme_quadrant_level0 (:2717-2760, 2872-2920) */
            if (single) {
                w  = c->L.hme_tw0;
                h  = c->L.hme_th0;
                ox = (int16_t)(-(int16_t)(w >> 1) + (int16_t)(xsc >> 2));
                oy = (int16_t)(-(int16_t)(h >> 1) + (int16_t)(ysc >> 2));
            } else {
                w = c->L.hme_w0[rw];
                h = c->L.hme_h0[rh];
                int16_t ddx = (int16_t)(xsc >> 2), ddy = (int16_t)(ysc >> 2);
                if (rw > 0) ddx = (int16_t)(ddx + c->L.hme_w0[0]);
                if (rh > 0) ddy = (int16_t)(ddy + c->L.hme_h0[0]);
                ox = (int16_t)(-(int16_t)(c->L.hme_tw0 >> 1) + ddx);
                oy = (int16_t)(-(int16_t)(c->L.hme_th0 >> 1) + ddy);
            }
        } else if (lvl == 1) {
            w  = me_hme_round_w((int16_t)p->hme_level1_search_area_in_width_array[rw]);
            h  = (int16_t)p->hme_level1_search_area_in_height_array[rh];
            ox = (int16_t)(-(w >> 1) + (int16_t)((int16_t)ME_PLAN_RD(st->hme_x[0][k]) >> 1));
            oy = (int16_t)(-(h >> 1) + (int16_t)((int16_t)ME_PLAN_RD(st->hme_y[0][k]) >> 1));
        } else {
            w  = me_hme_round_w((int16_t)p->hme_level2_search_area_in_width_array[rw]);
            h  = (int16_t)p->hme_level2_search_area_in_height_array[rh];
            ox = (int16_t)(-(w >> 1) + (int16_t)ME_PLAN_RD(st->hme_x[1][k]));
            oy = (int16_t)(-(h >> 1) + (int16_t)ME_PLAN_RD(st->hme_y[1][k]));
        }
        me_clip_area(g.ox, &ox, &w, g.pad_w, g.ref_w);
        me_clip_area(g.oy, &oy, &h, g.pad_h, g.ref_h);
        if (single && (w & 15) != 0) w = (int16_t)((w >> 4) << 4);
        st->hme_cox[k] = ox; st->hme_coy[k] = oy; /* kept even when nothing is searched: the centre still moves by them */
        if (w <= 0 || h <= 0) continue;
        st->hme_cw[k] = w; st->hme_ch[k] = h;
        const int wbytes = w + g.bw + 3;
        int       ws     = ((wbytes + 3) & ~3) + 4;
        if (((ws >> 2) & 1) == 0) ws += 4;
        const int ng = (g.bw & 3) == 0 ? (w + 3) >> 2 : w;
        for (int y = 0; y < h && ne < ME_HME_MAX_WIN;) {
            /* search rows that still fit the scratch: usually all of them -- the division only runs otherwise */
            int nr = h - y;
            if (ws * (nr + span) > c->hme_scratch_bytes - bytes) nr = (c->hme_scratch_bytes - bytes) / ws - span;
            if (nr < 1 && bytes > 0) { /* close the batch and retry with an empty scratch */
                st->hme_bstart[++nb] = ne; bytes = 0; tl = 0; ts = 0;
                continue;
            }
            if (nr < 1) break; /* cannot happen with the scratch sizes of me_lds_layout_compute */
            me_hme_win *wn = &st->hme_win[ne++];
            wn->off = bytes; wn->wstride = ws; wn->nd = (wbytes + 3) >> 2; wn->rows = nr + span; wn->sw = w; wn->sh = nr;
            wn->gx = g.ox + ox; wn->gy = g.oy + oy + y; wn->slot = k; wn->y0 = y; wn->tl = tl; wn->ts = ts;
            wn->inv_nu = me_magic_small(ME_HME_UNITS(wn->nd)); wn->inv_ng = me_magic_small(ng);
            tl += ME_HME_UNITS(wn->nd) * wn->rows; ts += ng * nr;
            bytes += ws * (nr + span); y += nr;
        }
    }
    if (ne > st->hme_bstart[nb]) st->hme_bstart[++nb] = ne;
    st->hme_nbatch = nb;
}

/* results of one HME level (run by ONE thread): position scaling and SAD*2 as in hme_level0/1/2 */
SVT_DEV void me_hme_finish_level(const me_ctx_t *c, int lvl) {
    const svt_me_params *p  = c->p;
    me_state_t          *st = c->st;
    const int            NW = p->number_hme_search_region_in_width, NH = p->number_hme_search_region_in_height;
    const int single = lvl == 0 && p->single_hme_quadrant && !p->enable_hme_level_1_flag && !p->enable_hme_level_2_flag;
    const int scale  = 4 >> lvl;
    for (int k = 0; k < 4; k++) {
        if (single ? k != 0 : ((k & 1) >= NW || (k >> 1) >= NH)) continue;
        uint64_t sad = 0xffffff;
        int16_t  x = (int16_t)ME_PLAN_RD(st->hme_x[lvl][k]), y = (int16_t)ME_PLAN_RD(st->hme_y[lvl][k]);
        if (st->hme_cw[k] > 0) {
            const uint64_t key = st->hme_keys[k];
            if (key != ~0ull) {
                const uint32_t idx = (uint32_t)key, sd = (uint32_t)(key >> 32);
                if (sd < sad) { sad = sd; x = (int16_t)(idx & 0xffffu); y = (int16_t)(idx >> 16); }
            }
        }
        st->hme_sad[lvl][k] = sad * 2;
        x = (int16_t)(x + st->hme_cox[k]); x = (int16_t)(x * scale);
        y = (int16_t)(y + st->hme_coy[k]); y = (int16_t)(y * scale);
        st->hme_x[lvl][k] = x; st->hme_y[lvl][k] = y;
    }
}

/* pick the search centre from the last enabled level (run by ONE thread), Codec/EbMotionEstimation.c:4880-4980 */
SVT_DEV void me_hme_select(const me_ctx_t *c, int list) {
    const svt_me_params *p  = c->p;
    me_state_t          *st = c->st;
    const int            NW = p->number_hme_search_region_in_width, NH = p->number_hme_search_region_in_height;
    const int            lvl = p->enable_hme_level_2_flag ? 2 : p->enable_hme_level_1_flag ? 1 : 0;
    if (!p->enable_hme_level_0_flag && lvl == 0) return; /* no level ran: the previous result stays */
    int16_t  xc = st->hme_x[lvl][0], yc = st->hme_y[lvl][0];
    uint64_t sd = st->hme_sad[lvl][0];
    if (!(lvl == 0 && p->single_hme_quadrant)) {
        for (int k = 1; k < 4; k++) {
            if ((k & 1) >= NW || (k >> 1) >= NH) continue;
            if (st->hme_sad[lvl][k] < sd) { xc = st->hme_x[lvl][k]; yc = st->hme_y[lvl][k]; sd = st->hme_sad[lvl][k]; }
        }
        st->hme_rh = NH;
    }
    if (lvl == 2) {
        /* [quirk] the reference sorts with the index pair (q / NW, q % NW) applied to its [rw][rh] arrays (:4943-4975):
         * element q is region rw = q / NW, rh = q % NW, i.e. slot (q % NW) * 2 + q / NW */
        const int tot = NH * NW;
        if (p->same_ref_poc && list == 1 && tot > 1) {
            for (int q = 0; q < tot - 1; q++)
                for (int n = q + 1; n < tot; n++) {
                    const int kq = (q % NW) * 2 + q / NW, kn = (n % NW) * 2 + n / NW;
                    if (st->hme_sad[2][kq] > st->hme_sad[2][kn]) {
                        const int16_t  tx = st->hme_x[2][kq], ty = st->hme_y[2][kq];
                        const uint64_t td = st->hme_sad[2][kq];
                        st->hme_x[2][kq] = st->hme_x[2][kn]; st->hme_y[2][kq] = st->hme_y[2][kn]; st->hme_sad[2][kq] = st->hme_sad[2][kn];
                        st->hme_x[2][kn] = tx; st->hme_y[2][kn] = ty; st->hme_sad[2][kn] = td;
                    }
                }
            xc = st->hme_x[2][2]; yc = st->hme_y[2][2]; /* element [0][1] of the reference's arrays: rw = 0, rh = 1 */
        }
    }
    st->hme_xc = xc; st->hme_yc = yc;
}

#ifndef SVT_HOST_EMU
/* The single-thread sections between the phases of an HME level -- results of level `fin` (me_hme_finish_level), plan of level `plan`
 * (me_hme_plan_level); either may be -1 -- spread over lanes 0..3 of wave 0: lane k owns region slot k = rh * 2 + rw.  The regions are
 * independent up to the packing of their windows into the scratch (prefix sums over the four lanes); when the level's windows do not fit
 * the scratch together (the 64x64-area presets' level 0) lane 0 plans the level with the sequential code.  The serial chain of LDS round
 * trips and reciprocal loads of one lane was 17 - 21 % of a workgroup's time at 1080p / 360p (SVT_HIP_ME_PROFILE); same results. */
SVT_DEV uint32_t me_magic_lane(int d) { return d <= 256 ? me_magics.v[d] : me_magic_of(d); }
SVT_DEV void me_hme_lanes(const me_ctx_t *c, int tid, int list, int fin, int plan, int16_t xsc, int16_t ysc, int first) {
    if (tid >= 4) return;
    const svt_me_params *p  = c->p;
    me_state_t          *st = c->st;
    const int            NW = p->number_hme_search_region_in_width, NH = p->number_hme_search_region_in_height;
    const int            k = tid, rw = k & 1, rh = k >> 1;
    const bool           valid = rw < NW && rh < NH;
    if (fin >= 0 && valid) {
        const int scale = 4 >> fin;
        uint64_t  sad = 0xffffff;
        int16_t   x = st->hme_x[fin][k], y = st->hme_y[fin][k];
        if (st->hme_cw[k] > 0) {
            const uint64_t key = st->hme_keys[k];
            if (key != ~0ull) {
                const uint32_t idx = (uint32_t)key, sd = (uint32_t)(key >> 32);
                if (sd < sad) { sad = sd; x = (int16_t)(idx & 0xffffu); y = (int16_t)(idx >> 16); }
            }
        }
        st->hme_sad[fin][k] = sad * 2;
        x = (int16_t)(x + st->hme_cox[k]); x = (int16_t)(x * scale);
        y = (int16_t)(y + st->hme_coy[k]); y = (int16_t)(y * scale);
        st->hme_x[fin][k] = x; st->hme_y[fin][k] = y;
    }
    if (plan < 0) return;
    me_hme_geom g;
    me_hme_geom_of(c, list, plan, &g);
    const int span = 2 * (g.bh - 1);
    if (first) { /* [quirk] centres are only initialised while the reference's row counter is below NH */
        const int rh_old = st->hme_rh;
        if (rh_old < NH && valid && rh >= rh_old) {
            st->hme_x[0][k] = (int16_t)(xsc >> 2); st->hme_y[0][k] = (int16_t)(ysc >> 2);
            st->hme_x[1][k] = (int16_t)(xsc >> 1); st->hme_y[1][k] = (int16_t)(ysc >> 1);
            st->hme_x[2][k] = xsc; st->hme_y[2][k] = ysc;
        }
    }
    { uint32_t *kw_ = (uint32_t *)&st->hme_keys[k]; kw_[0] = ~0u; kw_[1] = ~0u; }
    int16_t w = 0, h = 0, ox = 0, oy = 0;
    if (valid) {
        if (plan == 0) {
            w = rw ? c->L.hme_w0[1] : c->L.hme_w0[0]; /* (selects: a lane-dependent index would move the whole structure to scratch memory) */
            h = rh ? c->L.hme_h0[1] : c->L.hme_h0[0];
            int16_t ddx = (int16_t)(xsc >> 2), ddy = (int16_t)(ysc >> 2);
            if (rw > 0) ddx = (int16_t)(ddx + c->L.hme_w0[0]);
            if (rh > 0) ddy = (int16_t)(ddy + c->L.hme_h0[0]);
            ox = (int16_t)(-(int16_t)(c->L.hme_tw0 >> 1) + ddx);
            oy = (int16_t)(-(int16_t)(c->L.hme_th0 >> 1) + ddy);
        } else if (plan == 1) {
            w  = me_hme_round_w((int16_t)(rw ? p->hme_level1_search_area_in_width_array[1] : p->hme_level1_search_area_in_width_array[0]));
            h  = (int16_t)(rh ? p->hme_level1_search_area_in_height_array[1] : p->hme_level1_search_area_in_height_array[0]);
            ox = (int16_t)(-(w >> 1) + (int16_t)(st->hme_x[0][k] >> 1));
            oy = (int16_t)(-(h >> 1) + (int16_t)(st->hme_y[0][k] >> 1));
        } else {
            w  = me_hme_round_w((int16_t)(rw ? p->hme_level2_search_area_in_width_array[1] : p->hme_level2_search_area_in_width_array[0]));
            h  = (int16_t)(rh ? p->hme_level2_search_area_in_height_array[1] : p->hme_level2_search_area_in_height_array[0]);
            ox = (int16_t)(-(w >> 1) + st->hme_x[1][k]);
            oy = (int16_t)(-(h >> 1) + st->hme_y[1][k]);
        }
        me_clip_area(g.ox, &ox, &w, g.pad_w, g.ref_w);
        me_clip_area(g.oy, &oy, &h, g.pad_h, g.ref_h);
    }
    const bool ok = valid && w > 0 && h > 0;
    st->hme_cox[k] = valid ? ox : (int16_t)0; st->hme_coy[k] = valid ? oy : (int16_t)0; /* kept even when nothing is searched: the centre still moves by them */
    st->hme_cw[k] = ok ? w : (int16_t)0; st->hme_ch[k] = ok ? h : (int16_t)0;
    const int wbytes = w + g.bw + 3;
    int       ws     = ((wbytes + 3) & ~3) + 4;
    if (((ws >> 2) & 1) == 0) ws += 4;
    const int nd = (wbytes + 3) >> 2, ng = (g.bw & 3) == 0 ? (w + 3) >> 2 : w;
    const int my_bytes = ok ? ws * (h + span) : 0, my_tl = ok ? ME_HME_UNITS(nd) * (h + span) : 0, my_ts = ok ? ng * h : 0;
    int       bytes = 0, tl = 0, ts = 0, ne = 0, total = 0, n_all = 0;
    _Pragma("unroll") for (int j = 0; j < 4; j++) {
        const int bj = __shfl(my_bytes, j), tlj = __shfl(my_tl, j), tsj = __shfl(my_ts, j), okj = __shfl((int)ok, j);
        if (j < k) { bytes += bj; tl += tlj; ts += tsj; ne += okj; }
        total += bj; n_all += okj;
    }
    if (total <= c->hme_scratch_bytes) { /* one batch, one window per region */
        if (ok) {
            me_hme_win *wn = &st->hme_win[ne];
            wn->off = (uint32_t)bytes; wn->wstride = (uint16_t)ws; wn->nd = (uint8_t)nd; wn->rows = (uint16_t)(h + span); wn->sw = (uint16_t)w; wn->sh = (uint16_t)h;
            wn->gx = (int16_t)(g.ox + ox); wn->gy = (int16_t)(g.oy + oy); wn->slot = (uint8_t)k; wn->y0 = 0; wn->tl = (uint16_t)tl; wn->ts = (uint16_t)ts;
            wn->inv_nu = me_magic_lane(ME_HME_UNITS(nd)); wn->inv_ng = me_magic_lane(ng);
        }
        if (k == 0) { st->hme_rh = NH; st->hme_bstart[0] = 0; st->hme_bstart[1] = n_all; st->hme_nbatch = n_all > 0 ? 1 : 0; }
    } else if (k == 0) me_hme_plan_level(c, list, plan, xsc, ysc, first); /* bands: the sequential planner (it repeats the steps above) */
}
#endif


#ifdef SVT_HOST_EMU
static inline
#else
__device__ __forceinline__
#endif
void me_sb_run(const me_ctx_t *c, int tid_) {
    int tid = tid_;
    (void)tid;
    const svt_me_params *p  = c->p;
    me_state_t          *st = c->st;
    const int            nlist = p->num_ref_lists;
    const int            NW = p->number_hme_search_region_in_width, NH = p->number_hme_search_region_in_height;
#ifdef SVT_HOST_EMU
#define ME_PRED0_REGS (c->pred0 + tid)
#else
    uint32_t pred0_regs[16]; /* list 0 prediction dwords of this lane's bi-pred work, see ph_store_pred0 */
#define ME_PRED0_REGS pred0_regs
#endif
    int16_t  xsc = 0, ysc = 0;
#ifndef SVT_HOST_EMU
    unsigned long long mark_t_ = c->prof ? __builtin_amdgcn_s_memtime() : 0;
#endif

    ME_PHASE(ph_init(c, tid));
    ME_MARK(0);

    for (int list = 0; list < nlist; list++) {
        /* the reference's plane descriptors are read many times (address arithmetic, clipping): keep them in LDS */
        ME_PHASE(if (tid < (int)(3 * sizeof(svt_plane) / 4)) ((uint32_t *)st->refd)[tid] = ((const uint32_t *)&c->pic->ref[list])[tid];
                 if (tid >= 64 && tid < 72) st->red[tid - 64] = 0;
                 /* compact layout: list 0's SSD tables have overwritten the quarter-resolution SB */
                 if (c->L.compact && list == 1 && p->enable_hme_level_1_flag && p->fractional_search_method == SVT_SSD_SEARCH) ph_load_quarter(c, tid));
        const svt_plane  rf_u = me_plane_uni(&st->refd[0]); /* full-resolution reference plane of this list */
        const svt_plane *rf = &rf_u;
        const int        ox = (int16_t)c->sb_x, oy = (int16_t)c->sb_y;
        uint64_t         zero_c = 0; /* 2 * SAD of the block at (0, 0) of this list, when test_search_area_bounds ran */
        int              have_zero = 0;
        if (p->temporal_layer_index > 0 || list == 0) {
            /* ---- test_search_area_bounds ---- */
            {
                const int pad = ME_SB - 1, W = rf->width, H = rf->height;
                const int tw = p->hme_level0_total_search_area_width, th = p->hme_level0_total_search_area_height;
                int16_t   dx[5], dy[5];
                dx[0] = 0; dy[0] = 0;
                dx[1] = me_clip_center(ox, (int16_t)tw, pad, W); dy[1] = me_clip_center(oy, 0, pad, H);
                dx[2] = me_clip_center(ox, 0, pad, W); dy[2] = me_clip_center(oy, (int16_t)(0 - th), pad, H);
                dx[3] = me_clip_center(ox, 0, pad, W); dy[3] = me_clip_center(oy, (int16_t)th, pad, H);
                int16_t dirx = 0, diry = 0;
                int     nc = 4;
                if (list == 1) {
                    const uint32_t mv00 = (uint32_t)ME_UNI(st->best_mv[0][0]);
                    dirx = (int16_t)(0 - (me_mvx(mv00) >> 2));
                    diry = (int16_t)(0 - (me_mvy(mv00) >> 2));
                    dx[4] = me_clip_center(ox, dirx, pad, W); dy[4] = me_clip_center(oy, diry, pad, H);
                    nc = 5;
                }
                ME_PHASE(ph_center_sads(c, tid, rf, nc, dx, dy)); /* red[] was zeroed with the plane descriptors above */
                zero_c = (uint64_t)(uint32_t)ME_UNI(st->red[0]) << 1; have_zero = 1;
                uint64_t b_c = (uint64_t)(uint32_t)ME_UNI(st->red[1]) << 1,
                         c_c = (uint64_t)(uint32_t)ME_UNI(st->red[2]) << 1, d_c = (uint64_t)(uint32_t)ME_UNI(st->red[3]) << 1;
                uint64_t a_c = zero_c; /* [quirk] A is evaluated at the zero-MV address (:4302-4327) */
                uint64_t dir_c = list == 1 ? (uint64_t)(uint32_t)ME_UNI(st->red[4]) << 1 : 0xFFFFFFFFFFFFFull;
                uint64_t best = zero_c;
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
                /* red[] is zeroed again when the search region is staged: a barrier must lie between; the HME phases bring theirs */
                if (!(p->enable_hme_flag && c->sb_h == ME_SB && (p->enable_hme_level_0_flag || p->enable_hme_level_1_flag || p->enable_hme_level_2_flag)))
                    ME_PHASE((void)0);
            }
            ME_MARK(1);
            /* ---- HME ---- */
            if (p->enable_hme_flag && c->sb_h == ME_SB) {
                /* The control flow of the hierarchical search (area placement, clipping, batching, scaling, the choice
                 * between the regions) runs on one thread and lives in LDS; the 256 threads only execute the load and
                 * search phases of each batch of windows.  The three levels share one instance of that code. */
                const int last_lvl = p->enable_hme_level_2_flag ? 2 : p->enable_hme_level_1_flag ? 1 : 0;
                int       first = 1, planned = 0;
#define ME_HME_LEVEL_ON(l) ((l) == 0 ? p->enable_hme_level_0_flag : (l) == 1 ? p->enable_hme_level_1_flag : p->enable_hme_level_2_flag)
                for (int lvl = 0; lvl < 3; lvl++) {
                    if (!ME_HME_LEVEL_ON(lvl)) continue;
                    ME_SUBMARK_BEGIN();
                    /* (the plan of every level but the first rides with the previous level's finish: one single-thread section and one
                       barrier less per level) */
#ifndef SVT_HOST_EMU
                    const int lanes = !(p->single_hme_quadrant && !p->enable_hme_level_1_flag && !p->enable_hme_level_2_flag); /* (the one-region presets keep their short form) */
                    if (!planned && lanes) ME_PHASE(me_hme_lanes(c, tid, list, -1, lvl, xsc, ysc, first));
                    else
#endif
                    if (!planned) ME_UNIFORM_WRITE(me_hme_plan_level(c, list, lvl, xsc, ysc, first));
                    ME_SUBMARK(20);
                    ME_STOP_AT(19);
                    first = 0;
                    me_hme_geom g;
                    me_hme_geom_of(c, list, lvl, &g);
                    const int nbatch = ME_UNI(st->hme_nbatch);
                    for (int b = 0; b < nbatch; b++) {
                        const int         e0 = ME_UNI(st->hme_bstart[b]), e1 = ME_UNI(st->hme_bstart[b + 1]);
                        const me_hme_win *wl = &st->hme_win[e1 - 1];
                        const int         ntl = ME_UNI(wl->tl + ME_HME_UNITS(wl->nd) * wl->rows);
                        const int         nts = ME_UNI(wl->ts + ((g.bw & 3) == 0 ? (wl->sw + 3) >> 2 : wl->sw) * wl->sh);
                        int               slot_mask = 0;
                        for (int e = e0; e < e1; e++) slot_mask |= 1 << st->hme_win[e].slot;
                        slot_mask = ME_UNI(slot_mask);
                        ME_SUBMARK(22);
                        ME_PHASE(ph_hme_load_multi(c, tid, g.ref, st->hme_win, e0, e1, ntl));
                        ME_SUBMARK(14);
                        ME_STOP_AT(20);
                        ME_PHASE(ph_hme_search_multi(c, tid, g.blk, g.bstride, g.bw, g.bh, st->hme_win, e0, e1, nts, st->hme_keys, slot_mask));
                        ME_SUBMARK(15);
                        ME_STOP_AT(21);
                    }
                    ME_SUBMARK(22);
                    int nxt = -1;
                    for (int l2 = lvl + 1; l2 < 3 && nxt < 0; l2++) if (ME_HME_LEVEL_ON(l2)) nxt = l2;
#ifndef SVT_HOST_EMU
                    if (lanes) ME_PHASE(me_hme_lanes(c, tid, list, lvl, lvl == last_lvl ? -1 : nxt, xsc, ysc, 0); if (lvl == last_lvl && tid == 0) me_hme_select(c, list));
                    else
#endif
                    ME_UNIFORM_WRITE(me_hme_finish_level(c, lvl); if (lvl == last_lvl) me_hme_select(c, list); else if (nxt >= 0) me_hme_plan_level(c, list, nxt, xsc, ysc, 0));
                    planned = nxt >= 0 && lvl != last_lvl;
                    ME_SUBMARK(21);
                }
#undef ME_HME_LEVEL_ON
                xsc = (int16_t)ME_UNI(st->hme_xc); ysc = (int16_t)ME_UNI(st->hme_yc);
            }
            ME_MARK(2);
        } else {
            xsc = 0; ysc = 0;
        }

        int16_t saw = (int16_t)(p->search_area_width < 127 ? p->search_area_width : 127);
        int16_t sah = (int16_t)(p->search_area_height < 127 ? p->search_area_height : 127);
        int16_t sox, soy;
        int     W, H, w8, tail_extra, loaded = 0;
        /* search area of a centre: position, clipping, derived sizes (Codec/EbMotionEstimation.c:5008-5060) */
#define ME_SET_AREA()                                                                                               \
    do {                                                                                                            \
        saw = (int16_t)(p->search_area_width < 127 ? p->search_area_width : 127);                                    \
        sah = (int16_t)(p->search_area_height < 127 ? p->search_area_height : 127);                                  \
        sox = (int16_t)(xsc - (saw >> 1)); soy = (int16_t)(ysc - (sah >> 1));                                        \
        me_clip_area(ox, &sox, &saw, ME_SB - 1, c->pic_w);                                                           \
        me_clip_area(oy, &soy, &sah, ME_SB - 1, c->pic_h);                                                           \
        W = saw + ME_SB - 1; H = sah + ME_SB - 1; w8 = saw - (saw & 7); tail_extra = (saw & 7) ? 16 : 0;             \
        if (c->L.compact && W + ME_RGN_GX + 4 + tail_extra > c->L.region_stride) { /* no room for the tail columns: the launch with the full layout takes this SB */ \
            if (tid == 0) *SVT_AS_GLOBAL(uint32_t, c->redo) = 1;                                                      \
            return;                                                                                                  \
        }                                                                                                            \
    } while (0)
        /* stage the search region (+ halo) of this list in LDS; the full-pel keys are reset on the way */
#define ME_LOAD_REGION()                                                                                            \
    ME_PHASE(ph_load_rect(tid, c->region, c->L.region_stride, me_pix(rf, c->sb_x + sox - ME_RGN_GX, c->sb_y + soy - ME_RGN_GY), \
                          rf->stride, W + ME_RGN_GX + 4 + tail_extra, H + 2 * ME_RGN_GY + 1);                        \
             for (int t = tid; t < 85; t += SVT_NT) st->key[t] = ((uint64_t)ME_MAX_SAD_VALUE << 32);                  \
             if (tid < 8) st->red[tid] = 0;                                                                          \
             if (tid >= 128 && tid < 137) st->supel[tid - 128] = 0)
        if (xsc != 0 || ysc != 0) {
            /* ---- check_zero_zero_center (:4420-4500): the centre found above against (0, 0).  The SAD at (0, 0) is the
             * one test_search_area_bounds already computed for this list (same function, same block); the SAD at the
             * centre is taken from the search region, which is staged around that centre first -- when the centre
             * wins (the usual case) the region is already in place and no separate global pass is needed ---- */
            xsc = me_clip_center(ox, xsc, ME_SB - 1, rf->width);
            ysc = me_clip_center(oy, ysc, ME_SB - 1, rf->height);
            ME_SET_AREA();
            const int col = ME_RGN_GX + (xsc - sox), row = ME_RGN_GY + (ysc - soy);
            const int inside = have_zero && col >= 0 && row >= 0 && col + c->sb_w <= W + ME_RGN_GX + 4 + tail_extra &&
                               row + c->sb_h <= H + 2 * ME_RGN_GY + 1;
            uint64_t z, h;
            if (inside) {
                ME_LOAD_REGION();
                ME_PHASE(ph_region_center_sad(c, tid, col, row));
                z = zero_c; h = (uint64_t)(uint32_t)ME_UNI(st->red[1]) << 1;
                loaded = 1;
            } else {
                int16_t dx[2], dy[2];
                dx[0] = 0; dy[0] = 0; dx[1] = xsc; dy[1] = ysc;
                ME_PHASE(if (tid < 8) st->red[tid] = 0);
                ME_PHASE(ph_center_sads(c, tid, rf, 2, dx, dy));
                z = (uint64_t)(uint32_t)ME_UNI(st->red[0]) << 1; h = (uint64_t)(uint32_t)ME_UNI(st->red[1]) << 1;
            }
            uint64_t m = z < h ? z : h;
            if (m == z) { xsc = 0; ysc = 0; loaded = 0; }
            if (!loaded) ME_PHASE((void)0); /* everyone has read red[] before the staging below zeroes it */
        }
        ME_MARK(3);
        if (!loaded) {
            ME_SET_AREA();
            ME_LOAD_REGION();
        }
#undef ME_SET_AREA
#undef ME_LOAD_REGION

        ME_MARK(4);
        /* ---- full-pel search, in chunks of search rows ---- */
        {
            uint32_t *U = (uint32_t *)c->planes;
            if (saw >= 8 && (saw & 7) == 0 && saw * sah <= 4096) {
                ME_PHASE(ph_fullpel_fused(c, tid, saw, sah, ME_FULLPEL_UNROLL2(c)));
                ME_MARK(5);
                ME_MARK(6);
            } else {
            int max_pos   = c->L.scratch_bytes / (4 * ME_PU_STRIDE);
            int rows_chunk = max_pos / saw;
            if (rows_chunk < 1) rows_chunk = 1;
            if (rows_chunk > sah) rows_chunk = sah;
            for (int y0 = 0; y0 < sah; y0 += rows_chunk) {
                int ny = y0 + rows_chunk <= sah ? rows_chunk : sah - y0;
                ME_PHASE(ph_fullpel_sad8(c, tid, U, saw, y0, ny, w8));
                ME_PHASE(ph_fullpel_sum16(c, tid, U, saw, ny, w8));
                ME_PHASE(ph_fullpel_sum32(c, tid, U, ny * saw));
                ME_MARK(5);
                ME_PHASE(ph_fullpel_argmin(c, tid, U, saw, y0, ny));
                ME_MARK(6);
            }
            }
            /* keys -> best sad / mv (curr_mv = (y << 18) | (uint16)(x << 2), :108-110) and, where the sub-pel search is gated
             * (su_pel_enable :3839-4258: average MV magnitude / SAD per size class), the nine sums of that decision from the
             * values the lanes have just produced: one PU per lane, wave reductions (st->supel was zeroed with the keys) */
            ME_PHASE(if (tid < 128) {
                const int t = tid;
                uint32_t  mv = 0, sd = 0;
                if (t < 85) {
                    const uint64_t k = st->key[t];
                    const uint32_t idx = (uint32_t)k;
                    if (t == 0 && list == 1) c->cand[0] = 0; /* (after the read: the table may share the keys' bytes) bi-prediction sum of the 64x64 PU */
                    sd = (uint32_t)(k >> 32);
                    st->best_sad[list][t] = sd;
                    if (sd != (uint32_t)ME_MAX_SAD_VALUE) {
                        const int yq = me_udiv((int)idx, saw);
                        int       xi = (int)idx - yq * saw + sox, yi = yq + soy;
                        mv = (((uint32_t)(uint16_t)yi) << 18) | (uint16_t)((uint16_t)xi << 2);
                        st->best_mv[list][t] = mv;
                    } else mv = st->best_mv[list][t];
                }
                if (p->fractional_search_model == 1) {
                    const int cls = t >= 85 ? -1 : t >= 21 ? 2 : t >= 5 ? 1 : t >= 1 ? 0 : -1;
                    _Pragma("unroll") for (int k = 0; k < 3; k++) {
                        svt_wave_add_u32(&st->supel[3 * k + 0], cls == k ? (uint32_t)(int32_t)me_mvx(mv) : 0u, 1);
                        svt_wave_add_u32(&st->supel[3 * k + 1], cls == k ? (uint32_t)(int32_t)me_mvy(mv) : 0u, 1);
                        svt_wave_add_u32(&st->supel[3 * k + 2], cls == k ? sd : 0u, 1);
                    }
                }
            });
        }

        ME_MARK(7);
        /* ---- sub-pel ---- */
        int en32 = 0, en16 = 0, en8 = 0, enq = 0;
        if (p->fractional_search_model == 0) { en32 = en16 = en8 = enq = 1; }
        else if (p->fractional_search_model == 1) {
            int      sx = ME_UNI(st->supel[0]), sy = ME_UNI(st->supel[1]);
            uint32_t ss = (uint32_t)ME_UNI(st->supel[2]);
            uint32_t ax = (uint32_t)(sx >> 2), ay = (uint32_t)(sy >> 2);
            uint32_t mag32 = ax * ax + ay * ay, sad32 = ss >> 2;
            sx = ME_UNI(st->supel[3]); sy = ME_UNI(st->supel[4]); ss = (uint32_t)ME_UNI(st->supel[5]);
            ax = (uint32_t)(sx >> 4); ay = (uint32_t)(sy >> 4);
            uint32_t mag16 = ax * ax + ay * ay, sad16 = ss >> 4;
            sx = ME_UNI(st->supel[6]); sy = ME_UNI(st->supel[7]); ss = (uint32_t)ME_UNI(st->supel[8]);
            ax = (uint32_t)(sx >> 6); ay = (uint32_t)(sy >> 6);
            uint32_t mag8 = ax * ax + ay * ay, sad8 = ss >> 6;
            const int thr_[4]    = {48, 32, 80, 48};
            const int t32_[4][4] = {{1, 0, 1, 0}, {1, 0, 1, 1}, {1, 0, 1, 0}, {1, 1, 1, 0}};
            const int t16_[4]    = {0, 1, 0, 1};
            const int t8_[4][4]  = {{0, 1, 0, 1}, {0, 1, 0, 1}, {0, 1, 0, 1}, {0, 1, 0, 0}};
            int       tl = p->temporal_layer_index > 3 ? 3 : p->temporal_layer_index;
            uint32_t  t2 = (uint32_t)(thr_[tl] * thr_[tl]);
            en32 = t32_[tl][2 * !(mag32 < t2) + !(sad32 < 32 * 32 * 6)];
            en16 = t16_[2 * !(mag16 < t2) + !(sad16 < 16 * 16 * 2)];
            en8  = t8_[tl][2 * !(mag8 < t2) + !(sad8 < 8 * 8 * 2)];
            enq  = 1;
        }
        const int need_planes = en32 || en16 || en8 || enq || (nlist == 2);
        ME_MARK(8);
        if (need_planes) {
            ME_PHASE(ph_interp_strips(c, tid, W, H));
        }
        ME_MARK(9);
#ifndef SVT_HOST_EMU
        /* SUB_SAD refinement of the 32x32 / 16x16 PUs only (the M5+ presets): one phase, see ph_subpel_fast */
        /* with cu8x8_mode == 1 that holds for both lists whatever the gating says: the 32x32 / 16x16 levels of the bi-prediction
         * ride on the same lanes (fast_bi), the 64x64 level stays with ph_store_pred0 / ph_bipred */
        const int fast_bi = nlist == 2 && p->fractional_search_model != 2 && p->fractional_search_method == SVT_SUB_SAD_SEARCH &&
                            !p->fractional_search64x64 && p->cu16x16_mode == 0 && p->cu8x8_mode == 1;
        if (enq && p->fractional_search_method == SVT_SUB_SAD_SEARCH && !p->fractional_search64x64 && p->cu16x16_mode == 0 &&
            !(en8 && p->cu8x8_mode != 1)) {
            if (en32 || en16 || fast_bi) ME_PHASE(ph_subpel_fast(c, tid, list, sox, soy, en32, en16, fast_bi, ME_PRED0_REGS));
            ME_MARK(10);
            ME_MARK(11);
        } else
#endif
        if (en32 || en16 || en8 || enq) {
            ME_PHASE(me_cand_zero(c, tid, 0);
                     ph_subpel_prep(c, tid, en32, en16, en8));
            ME_PHASE(ph_halfpel(c, tid, list, sox, soy, en32, en16, en8));
            ME_PHASE(ph_halfpel_decide(c, tid, list, en32, en16, en8));
            ME_MARK(10);
            if (enq) {
                ME_PHASE(me_cand_zero(c, tid, 1));
                ME_PHASE(ph_quarterpel(c, tid, list, sox, soy, en32, en16, en8));
                ME_PHASE(ph_quarterpel_decide(c, tid, list, en32, en16, en8));
            }
            ME_MARK(11);
        }
        if (nlist == 2) {
#ifdef SVT_HOST_EMU
            const int lmax = 4, czero = 85;
#else
            const int lmax = fast_bi ? 0 : 4, czero = 85; /* fast_bi: every level is done, cand[0..20] hold the sums */
#endif
            if (lmax == 0) { /* nothing left */ }
            else if (list == 0) ME_PHASE(ph_store_pred0(c, tid, sox, soy, ME_PRED0_REGS, lmax));
            else {
                ME_PHASE(for (int t = tid; t < czero; t += SVT_NT) c->cand[t] = 0);
                ME_PHASE(ph_bipred(c, tid, sox, soy, ME_PRED0_REGS, lmax));
            }
            ME_MARK(12);
        }
    }

    /* ---- results ---- */
    uint32_t *ow = (uint32_t *)c->planes;
    ME_PHASE(ph_output(c, tid, 0, ow));
    {
        uint32_t SVT_GLOBAL *g = SVT_AS_GLOBAL(uint32_t, c->pic->results + (size_t)c->sb_index * 85);
#ifdef SVT_HOST_EMU
        for (int t = 0; t < 850; t++) g[t] = ow[t];
#else
        for (int t = tid; t < 850; t += SVT_NT) g[t] = ow[t];
#endif
        if (c->pic->rcme && tid == 0) {
            uint32_t acc = 0;
            for (int i = 0; i < 16; i++) acc += ow[(5 + i) * 10 + 2];
            *SVT_AS_GLOBAL(uint32_t, &c->pic->rcme[c->sb_index]) = acc;
        }
    }
    ME_MARK(13);
}

#endif /* SVT_ME_CORE_H */
