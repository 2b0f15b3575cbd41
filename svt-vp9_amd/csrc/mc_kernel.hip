/*
 * mc_kernel.hip -- inter prediction (VP9 regular 8-tap motion compensation) of whole pictures from their mode-info grid
 * (gfx950).
 *
 * Replaces, per picture, every call prediction_fun_table[0] = inter_prediction makes (Codec/EbEncDecProcess.c:132,
 * :277, :3800; Codec/EbIntraPrediction.c:49-72): build_inter_predictors (VPX/vp9_reconinter.c:102-252) with
 * eb_vp9_clamp_mv_to_umv_border_sb (:72-92) and the eb_vp9_convolve8* / copy / avg kernels inter_predictor picks
 * (VPX/vp9_reconinter.h:23-28, VPX/vp9_scale.c:76-84,126-128, VPX/vpx_convolve.c:20-215).
 *
 * Mapping.  The work unit is a 4-sample-wide tile of one plane, NR = 8 rows high in luma (half an 8x8 unit) and 4 in
 * chroma: a lane loads NR + 7 rows of 16 bytes around its tile, filters them horizontally into NR + 7 packed rows
 * (uint8, rounded and clipped exactly like the reference's intermediate buffer), filters those vertically into NR rows
 * and writes NR dwords.  Everything stays in registers -- no LDS, no barrier, no cross-lane traffic -- because
 * neighbouring 8x8 units may belong to blocks with unrelated motion vectors.  A workgroup of 128 lanes covers 32 units
 * of one mode-info row: wave 0 the luma tiles (64 consecutive dwords of a sample row), wave 1 the Cb and Cr tiles, so
 * the loads and stores of a wave are contiguous.  Workgroups are numbered so that each XCD works on one contiguous band
 * of mode-info rows: the rows above/below a tile that the 8-tap filter needs are then found in that XCD's own L2.
 * Filter phase 0 of the VP9 kernel is {0,0,0,128,0,0,0,0}: (128 p + 64) >> 7 == p, so the copy / horizontal-only /
 * vertical-only variants of the reference are the same computation with that phase and need no branches; phase 0 is
 * taken by a select because its tap 128 does not fit the signed-byte dot product.
 * Arithmetic: v_dot4_i32_i8 on samples biased by -128 (sum taps = 128 -> +128*128 restores it), integer throughout.
 * HBM traffic per picture: the reference planes once (tiles of neighbouring lanes overlap in L1/L2) + 1.5 W H written.
 */
#include <hip/hip_runtime.h>
#include "svt_ctx.h"

namespace {

struct mc_pic_dev {
    const svt_mc_mode_info *mi;
    int32_t        mi_stride, mi_rows, mi_cols;
    svt_yuv_planes ref[2];
    svt_yuv_planes pred;
    int32_t        use_subpel;
};

/* taps 0..3 and 4..7 of the 16 phases packed as signed bytes (VP9 regular filter, VPX/vp9_filter.c:32-47); phase 0 is
 * never used through this table */
#define PK(a, b, c, d) ((uint32_t)(uint8_t)(int8_t)(a) | (uint32_t)(uint8_t)(int8_t)(b) << 8 | (uint32_t)(uint8_t)(int8_t)(c) << 16 | (uint32_t)(uint8_t)(int8_t)(d) << 24)
__constant__ uint32_t c_tap_lo[16] = {PK(0, 0, 0, 0), PK(0, 1, -5, 126), PK(-1, 3, -10, 122), PK(-1, 4, -13, 118), PK(-1, 4, -16, 112), PK(-1, 5, -18, 105),
                                      PK(-1, 5, -19, 97), PK(-1, 6, -19, 88), PK(-1, 6, -19, 78), PK(-1, 5, -18, 68), PK(-1, 5, -16, 58), PK(-1, 4, -14, 48),
                                      PK(-1, 4, -11, 37), PK(-1, 3, -9, 27), PK(0, 2, -6, 18), PK(0, 1, -3, 8)};
__constant__ uint32_t c_tap_hi[16] = {PK(0, 0, 0, 0), PK(8, -3, 1, 0), PK(18, -6, 2, 0), PK(27, -9, 3, -1), PK(37, -11, 4, -1), PK(48, -14, 4, -1),
                                      PK(58, -16, 5, -1), PK(68, -18, 5, -1), PK(78, -19, 6, -1), PK(88, -19, 6, -1), PK(97, -19, 5, -1), PK(105, -18, 5, -1),
                                      PK(112, -16, 4, -1), PK(118, -13, 4, -1), PK(122, -10, 3, -1), PK(126, -5, 1, 0)};

typedef uint32_t u32x4a4 __attribute__((ext_vector_type(4), aligned(4)));

__device__ __forceinline__ uint32_t alignbyte(uint32_t hi, uint32_t lo, uint32_t sh) { return __builtin_amdgcn_alignbyte(hi, lo, sh); }
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : v > hi ? hi : v; }
/* ROUND_POWER_OF_TWO(sum, 7) then clip_pixel; `acc` already holds the +64 and the bias correction */
__device__ __forceinline__ uint32_t finish(int acc) { return (uint32_t)clampi(acc >> 7, 0, 255); }
/* first dot product of a chain: acc = dot4(a, b) + bias with the bias as a scalar operand.  Written as the VOP3 instruction: the
 * compiler otherwise picks the accumulate-in-place form (v_dot4c) and copies the bias into a fresh register for every chain --
 * one instruction in seven of this kernel */
__device__ __forceinline__ int dot4_bias(uint32_t a, uint32_t b, int bias) {
    int r;
    __asm__("v_dot4_i32_i8 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(bias));
    return r;
}
/* four clipped samples into a dword.  The halves are joined with an explicit v_perm_b32: written as shifts and ors, this
 * hipcc folds the clamps of samples 2 and 3 into the packing and produces wrong bytes (seen on gfx950, ROCm 7.2; the ME
 * kernel's interpolation hit the same fold) */
__device__ __forceinline__ uint32_t pack4(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    return __builtin_amdgcn_perm(c | (d << 8), a | (b << 8), 0x05040100u);
}

/* picture planes, mode info and prediction live in device memory: say so, and the loads/stores are global_* with no
 * aperture check instead of flat_* */
#define MC_GLOBAL __attribute__((address_space(1)))
#define MC_AS_GLOBAL(T, p) ((T MC_GLOBAL *)(uintptr_t)(p))

/* 4 x NR tile at plane position (x, y) displaced by (s_col, s_row) sixteenths, VP9 regular filter */
template <int NR>
__device__ __forceinline__ void mc_tile(const uint8_t *plane, int stride, int x, int y, int s_row, int s_col, int sx, int sy, uint32_t out[NR]) {
    constexpr int NM = NR + 7, NG = (NM + 3) / 4;
    const uint8_t *p0 = plane + (ptrdiff_t)(y + (s_row >> 4) - 3) * stride + (x + (s_col >> 4) - 3);
    const uint32_t sh = (uint32_t)((uintptr_t)p0 & 3);
    p0 -= sh;
    const uint32_t tl = c_tap_lo[sx], th = c_tap_hi[sx];
    uint32_t       mid[NM];
    /* wave-uniform shortcuts (whole waves of zero / full-sample motion are the common case in real pictures): without a
     * vertical phase in the wave the 7 extra rows are neither fetched nor filtered, without a horizontal phase the taps
     * are skipped.  Branches on ballots: every lane of the wave takes the same side. */
    const bool any_sx = __builtin_amdgcn_ballot_w64(sx != 0) != 0, any_sy = __builtin_amdgcn_ballot_w64(sy != 0) != 0;
    /* all row loads first (up to 15 x 16 bytes in flight per lane): the kernel is bound by the latency of these loads, not by
     * their bandwidth or by the filter arithmetic */
    u32x4a4 rows[NM];
    _Pragma("unroll") for (int r = 0; r < NM; r++)
        if (any_sy || (r >= 3 && r < 3 + NR)) rows[r] = *MC_AS_GLOBAL(const u32x4a4, p0 + (ptrdiff_t)r * stride);
    _Pragma("unroll") for (int r = 0; r < NM; r++) {
        if (!any_sy && (r < 3 || r >= 3 + NR)) { mid[r] = 0; continue; }
        const u32x4a4 d = rows[r];
        /* bytes 0..11 of the row window (sample x - 3 first) */
        const uint32_t e0 = alignbyte(d.y, d.x, sh), e1 = alignbyte(d.z, d.y, sh), e2 = alignbyte(d.w, d.z, sh);
        if (!any_sx) { mid[r] = alignbyte(e1, e0, 3); continue; }
        const uint32_t b0 = e0 ^ 0x80808080u, b1 = e1 ^ 0x80808080u, b2 = e2 ^ 0x80808080u;
        const int      bias = 128 * 128 + 64;
        int            a0 = __builtin_amdgcn_sdot4((int)b1, (int)th, dot4_bias(b0, tl, bias), false);
        int            a1 = __builtin_amdgcn_sdot4((int)alignbyte(b2, b1, 1), (int)th, dot4_bias(alignbyte(b1, b0, 1), tl, bias), false);
        int            a2 = __builtin_amdgcn_sdot4((int)alignbyte(b2, b1, 2), (int)th, dot4_bias(alignbyte(b1, b0, 2), tl, bias), false);
        int            a3 = __builtin_amdgcn_sdot4((int)alignbyte(b2, b1, 3), (int)th, dot4_bias(alignbyte(b1, b0, 3), tl, bias), false);
        const uint32_t f = pack4(finish(a0), finish(a1), finish(a2), finish(a3));
        mid[r] = sx ? f : alignbyte(e1, e0, 3); /* phase 0: the samples themselves (bytes 3..6) */
    }
    if (!any_sy) {
        _Pragma("unroll") for (int yy = 0; yy < NR; yy++) out[yy] = mid[yy + 3];
        return;
    }
    /* columns of the NM intermediate rows as byte streams: col[j][g] = rows 4g .. 4g+3 of column j (row NM = padding) */
    uint32_t col[4][NG];
    _Pragma("unroll") for (int g = 0; g < NG; g++) {
        const uint32_t r0 = mid[4 * g] ^ 0x80808080u, r1 = mid[4 * g + 1] ^ 0x80808080u, r2 = mid[4 * g + 2] ^ 0x80808080u;
        const uint32_t r3 = 4 * g + 3 < NM ? mid[4 * g + 3] ^ 0x80808080u : 0u;
        const uint32_t a0 = __builtin_amdgcn_perm(r1, r0, 0x05010400u), a1 = __builtin_amdgcn_perm(r1, r0, 0x07030602u);
        const uint32_t q0 = __builtin_amdgcn_perm(r3, r2, 0x05010400u), q1 = __builtin_amdgcn_perm(r3, r2, 0x07030602u);
        col[0][g] = __builtin_amdgcn_perm(q0, a0, 0x05040100u); col[1][g] = __builtin_amdgcn_perm(q0, a0, 0x07060302u);
        col[2][g] = __builtin_amdgcn_perm(q1, a1, 0x05040100u); col[3][g] = __builtin_amdgcn_perm(q1, a1, 0x07060302u);
    }
    const uint32_t vl = c_tap_lo[sy], vh = c_tap_hi[sy];
    _Pragma("unroll") for (int yy = 0; yy < NR; yy++) {
        uint32_t o[4];
        const int g0 = yy >> 2, sft = yy & 3;
        _Pragma("unroll") for (int j = 0; j < 4; j++) {
            const uint32_t lo = sft ? alignbyte(col[j][g0 + 1], col[j][g0], sft) : col[j][g0], hi = sft ? alignbyte(col[j][g0 + 2], col[j][g0 + 1], sft) : col[j][g0 + 1];
            o[j] = finish(__builtin_amdgcn_sdot4((int)hi, (int)vh, dot4_bias(lo, vl, 128 * 128 + 64), false));
        }
        out[yy] = sy ? pack4(o[0], o[1], o[2], o[3]) : mid[yy + 3];
    }
}

template <int NR>
__device__ __forceinline__ void mc_unit(const mc_pic_dev &P, int plane, int mi_row, int mi_col, int x, int y) {
    const uint32_t MC_GLOBAL *mp = MC_AS_GLOBAL(const uint32_t, P.mi + (size_t)mi_row * P.mi_stride + mi_col);
    const uint32_t  m0 = mp[0], m1 = mp[1], m2 = mp[2];
    const int       rl0 = (int8_t)(m2 & 0xff), rl1 = (int8_t)((m2 >> 8) & 0xff), bw8 = (int)((m2 >> 16) & 0xff), bh8 = (int)(m2 >> 24);
    if (rl0 < 0 || bw8 < 1 || bh8 < 1) return;
    const int ss = plane ? 1 : 0;
    /* the block this unit belongs to (blocks are aligned to their size) and its distance to the picture edges in 1/8
     * luma samples (Codec/EbEncDecProcess.c:3708-3719) */
    const int bc = mi_col - mi_col % bw8, br = mi_row - mi_row % bh8;
    const int to_left = -(bc * 64), to_right = (P.mi_cols - bw8 - bc) * 64, to_top = -(br * 64), to_bottom = (P.mi_rows - bh8 - br) * 64;
    const int bw = (bw8 * 8) >> ss, bh = (bh8 * 8) >> ss;
    const int spel_left = (4 + bw) << 4, spel_right = spel_left - 16, spel_top = (4 + bh) << 4, spel_bottom = spel_top - 16;
    const int sc = 1 << (1 - ss);
    uint32_t  acc[NR];
    const int nref = rl1 >= 0 ? 2 : 1;
    for (int k = 0; k < nref; k++) {
        const svt_yuv_planes &R = P.ref[(k ? rl1 : rl0) ? 1 : 0];
        int mv_row = (int16_t)(k ? m0 >> 16 : m0 & 0xffff) * sc, mv_col = (int16_t)(k ? m1 >> 16 : m1 & 0xffff) * sc;
        mv_col = clampi(mv_col, to_left * sc - spel_left, to_right * sc + spel_right);
        mv_row = clampi(mv_row, to_top * sc - spel_top, to_bottom * sc + spel_bottom);
        int s_row, s_col, sx, sy;
        if (P.use_subpel) { s_row = mv_row; s_col = mv_col; sx = s_col & 15; sy = s_row & 15; }
        else if (plane) { s_row = (mv_row + 4) & ~7; s_col = (mv_col + 4) & ~7; sx = s_col & 7; sy = s_row & 7; } /* [quirk] vp9_reconinter.c:176-180 */
        else { s_row = (mv_row + 8) & ~15; s_col = (mv_col + 8) & ~15; sx = s_col & 15; sy = s_row & 15; }
        const uint8_t *pl = plane == 0 ? R.y : plane == 1 ? R.u : R.v;
        uint32_t       o[NR];
        mc_tile<NR>(pl, plane ? R.uv_stride : R.y_stride, x, y, s_row, s_col, sx, sy, o);
        _Pragma("unroll") for (int r = 0; r < NR; r++)
            acc[r] = k ? __builtin_amdgcn_lerp(acc[r], o[r], 0x01010101u) : o[r]; /* ROUND_POWER_OF_TWO(dst + p, 1) per byte */
    }
    uint8_t  *dp = plane == 0 ? P.pred.y : plane == 1 ? P.pred.u : P.pred.v;
    const int ds = plane ? P.pred.uv_stride : P.pred.y_stride;
    _Pragma("unroll") for (int r = 0; r < NR; r++) *MC_AS_GLOBAL(uint32_t, dp + (size_t)(y + r) * ds + x) = acc[r];
}

__global__ __launch_bounds__(128) void svt_mc_kernel(const mc_pic_dev *__restrict__ pics, int max_rows, int xblocks, int total, int chunk) {
    /* consecutive workgroup ids go to consecutive XCDs: id b works on item (b & 7) * chunk + (b >> 3), so that XCD k
     * owns the contiguous items [k * chunk, (k + 1) * chunk) = a band of mode-info rows */
    const int item = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3);
    if (item >= total) return;
    const int xb = item % xblocks, rowpic = item / xblocks, mi_row = rowpic % max_rows;
    const mc_pic_dev &P = pics[rowpic / max_rows];
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    if (mi_row >= P.mi_rows) return;
    if (wave == 0) { /* luma: unit lane >> 1, tile column lane & 1, 4 x 8 samples */
        const int mi_col = xb * 32 + (lane >> 1);
        if (mi_col < P.mi_cols) mc_unit<8>(P, 0, mi_row, mi_col, mi_col * 8 + 4 * (lane & 1), mi_row * 8);
    } else {         /* Cb (lanes 0..31) | Cr: 4 x 4 samples */
        const int mi_col = xb * 32 + (lane & 31);
        if (mi_col < P.mi_cols) mc_unit<4>(P, 1 + (lane >> 5), mi_row, mi_col, mi_col * 4, mi_row * 4);
    }
}

int mc_launch(svt_hip_ctx *ctx, int n_pics, const svt_mc_picture *pics) {
    int max_rows = 0, max_cols = 0;
    for (int i = 0; i < n_pics; i++) {
        const svt_mc_picture &p = pics[i];
        if (!p.d_mi || p.mi_rows < 1 || p.mi_cols < 1 || p.mi_stride < p.mi_cols || !p.pred.y || !p.pred.u || !p.pred.v)
            return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "mc: bad picture descriptor");
        for (int l = 0; l < 2; l++)
            if (!p.ref[l].y || !p.ref[l].u || !p.ref[l].v) return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "mc: both reference lists need planes");
        if (((uintptr_t)p.pred.y | (uintptr_t)p.pred.u | (uintptr_t)p.pred.v | (uintptr_t)p.pred.y_stride | (uintptr_t)p.pred.uv_stride) & 3)
            return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "mc: prediction planes and strides must be multiples of 4 bytes");
        max_rows = p.mi_rows > max_rows ? p.mi_rows : max_rows;
        max_cols = p.mi_cols > max_cols ? p.mi_cols : max_cols;
    }
    mc_pic_dev *h = nullptr, *d = nullptr;
    if (svt_ctx_stage(ctx, sizeof(mc_pic_dev) * (size_t)n_pics, (void **)&h, (void **)&d)) return svt_set_error(SVT_HIP_ERR_NO_RESOURCES, "mc: scratch");
    for (int i = 0; i < n_pics; i++) {
        h[i].mi = pics[i].d_mi; h[i].mi_stride = pics[i].mi_stride; h[i].mi_rows = pics[i].mi_rows; h[i].mi_cols = pics[i].mi_cols;
        h[i].ref[0] = pics[i].ref[0]; h[i].ref[1] = pics[i].ref[1]; h[i].pred = pics[i].pred; h[i].use_subpel = pics[i].use_subpel;
    }
    HIP_TRY(hipMemcpyAsync(d, h, sizeof(mc_pic_dev) * (size_t)n_pics, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipEventRecord(ctx->ev_start, ctx->stream));
    const int xblocks = (max_cols + 31) / 32, total = xblocks * max_rows * n_pics, chunk = (total + 7) / 8;
    hipLaunchKernelGGL(svt_mc_kernel, dim3(chunk * 8), dim3(128), 0, ctx->stream, (const mc_pic_dev *)d, max_rows, xblocks, total, chunk);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(ctx->ev_stop, ctx->stream));
    svt_ctx_stage_commit(ctx);
    ctx->timed = 1;
    return SVT_HIP_OK;
}
} // namespace

extern "C" int32_t svt_hip_inter_pred_batch_device(svt_hip_ctx *ctx, int32_t n_pics, const svt_mc_picture *pics) {
    if (!ctx || n_pics < 1 || !pics) return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "mc: null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    return mc_launch(ctx, n_pics, pics);
}

extern "C" int32_t svt_hip_inter_pred_frame(svt_hip_ctx *ctx, const svt_mc_mode_info *mi, int32_t mi_stride, int32_t mi_rows, int32_t mi_cols,
                                            const svt_mc_host_ref ref[2], int32_t use_subpel, uint8_t *pred_y, uint8_t *pred_u, uint8_t *pred_v) {
    if (!ctx || !mi || !ref || !pred_y || !pred_u || !pred_v || mi_rows < 1 || mi_cols < 1 || mi_stride < mi_cols)
        return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "mc: null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    const int      W = mi_cols * 8, H = mi_rows * 8;
    svt_mc_picture p;
    memset(&p, 0, sizeof p);
    const size_t mi_bytes = sizeof(svt_mc_mode_info) * (size_t)mi_rows * mi_stride;
    void        *dmi      = svt_ctx_slot(ctx, 26, mi_bytes);
    if (!dmi) return svt_set_error(SVT_HIP_ERR_NO_RESOURCES, "mc: device buffers");
    HIP_TRY(hipMemcpyAsync(dmi, mi, mi_bytes, hipMemcpyHostToDevice, ctx->stream));
    p.d_mi = (const svt_mc_mode_info *)dmi; p.mi_stride = mi_stride; p.mi_rows = mi_rows; p.mi_cols = mi_cols; p.use_subpel = use_subpel;
    for (int l = 0; l < 2; l++) {
        const svt_mc_host_ref &r = ref[l];
        if (!r.y || !r.u || !r.v || r.org_x < 80 || r.org_y < 80 || (r.org_x & 1) || (r.org_y & 1))
            return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "mc: reference planes need an even padding of at least 80 luma samples");
        const size_t ny = (size_t)r.y_stride * (H + 2 * r.org_y), nuv = (size_t)r.uv_stride * (H / 2 + r.org_y);
        uint8_t *dy = (uint8_t *)svt_ctx_slot(ctx, 27 + 3 * l, ny + 64);
        uint8_t *du = (uint8_t *)svt_ctx_slot(ctx, 28 + 3 * l, nuv + 64), *dv = (uint8_t *)svt_ctx_slot(ctx, 29 + 3 * l, nuv + 64);
        if (!dy || !du || !dv) return svt_set_error(SVT_HIP_ERR_NO_RESOURCES, "mc: device buffers");
        HIP_TRY(hipMemcpyAsync(dy, r.y, ny, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(hipMemcpyAsync(du, r.u, nuv, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(hipMemcpyAsync(dv, r.v, nuv, hipMemcpyHostToDevice, ctx->stream));
        p.ref[l].y = dy + (size_t)r.org_y * r.y_stride + r.org_x;
        p.ref[l].u = du + (size_t)(r.org_y / 2) * r.uv_stride + r.org_x / 2;
        p.ref[l].v = dv + (size_t)(r.org_y / 2) * r.uv_stride + r.org_x / 2;
        p.ref[l].y_stride = r.y_stride; p.ref[l].uv_stride = r.uv_stride; p.ref[l].width = W; p.ref[l].height = H;
    }
    /* prediction planes: one buffer, Y then U then V; units of non-inter blocks keep the caller's samples */
    const size_t ysz = (size_t)W * H, csz = ysz / 4;
    uint8_t     *dp  = (uint8_t *)svt_ctx_slot(ctx, 33, ysz + 2 * csz);
    if (!dp) return svt_set_error(SVT_HIP_ERR_NO_RESOURCES, "mc: device buffers");
    HIP_TRY(hipMemcpyAsync(dp, pred_y, ysz, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(dp + ysz, pred_u, csz, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(dp + ysz + csz, pred_v, csz, hipMemcpyHostToDevice, ctx->stream));
    p.pred.y = dp; p.pred.u = dp + ysz; p.pred.v = dp + ysz + csz; p.pred.y_stride = W; p.pred.uv_stride = W / 2; p.pred.width = W; p.pred.height = H;
    int32_t rc = mc_launch(ctx, 1, &p);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(pred_y, dp, ysz, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(pred_u, dp + ysz, csz, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(pred_v, dp + ysz + csz, csz, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return SVT_HIP_OK;
}
