/*
 * pa_kernel.hip -- picture-analysis pre-ME stage on gfx950: border replication and 1/4, 1/16 point decimation of the
 * input luma, producing the three padded planes motion estimation reads (an EbPaReferenceObject).
 *
 * Replaces pad_picture_to_multiple_of_sb_dimensions + decimate_input_picture of eb_vp9_picture_analysis_kernel
 * (Source/Lib/Codec/EbPictureAnalysisProcess.c:5010-5088 -> eb_vp9_decimation_2d :102-122, eb_vp9_generate_padding
 * Codec/EbMcp.c:17-58).  Pure streaming: every destination sample is src[step*clamp(y - pad)][step*clamp(x - pad)], so
 * one kernel writes all planes of all pictures of a batch; a thread produces 4 consecutive destination bytes.
 * HBM bound: reads W*H (the decimated planes re-read lines that the L2 still holds), writes ~1.4 W*H.
 */
#include <hip/hip_runtime.h>
#include "svt_ctx.h"

/* pictures are device memory: global_* loads/stores, no aperture check */
#define PA_GLOBAL __attribute__((address_space(1)))
#define PA_AS_GLOBAL(T, p) ((T PA_GLOBAL *)(uintptr_t)(p))

namespace {
struct pa_job {
    const uint8_t *src; /* luma sample (0,0) */
    uint8_t       *dst; /* first byte of the padded destination buffer */
    int32_t        sstride, dstride, step, pad_x, pad_y, dw, dh; /* dw x dh = decimated size (without padding) */
    int32_t        row0; /* first workgroup row of this job in the flattened grid */
};

constexpr int PA_ROWS = 4; /* destination rows per workgroup */

typedef uint32_t pa_u32x4 __attribute__((ext_vector_type(4), aligned(4)));
typedef uint32_t pa_u32x2 __attribute__((ext_vector_type(2), aligned(4)));

/* 4 destination bytes whose sources lie inside the picture: `s` = address of the first source byte, consecutive destination
 * bytes are `step` source bytes apart.  Global loads need no alignment on gfx950: one load per destination dword. */
__device__ __forceinline__ uint32_t pa_dword_inside(const uint8_t PA_GLOBAL *s, int step) {
    if (step == 1) return *(const uint32_t PA_GLOBAL *)s;
    if (step == 2) {
        const pa_u32x2 d = *(const pa_u32x2 PA_GLOBAL *)s;
        return __builtin_amdgcn_perm(d.y, d.x, 0x06040200u);
    }
    const pa_u32x4 d = *(const pa_u32x4 PA_GLOBAL *)s; /* step 4: bytes 0, 4, 8, 12 */
    return __builtin_amdgcn_perm(d.y, d.x, 0x0c0c0400u) | __builtin_amdgcn_perm(d.w, d.z, 0x04000c0cu);
}

__global__ __launch_bounds__(256) void svt_pa_plane_kernel(const pa_job *__restrict__ jobs, int n_jobs) {
    /* find the job of this workgroup row (a handful of jobs: linear scan) */
    int j = 0;
    while (j + 1 < n_jobs && (int)blockIdx.x >= jobs[j + 1].row0) j++;
    const pa_job J = jobs[j];
    const int tw = J.dw + 2 * J.pad_x, th = J.dh + 2 * J.pad_y; /* padded size */
    const int py0 = ((int)blockIdx.x - J.row0) * PA_ROWS;
    const int nu = (tw + 15) >> 4; /* 16-byte units per destination row: a thread produces one */
    for (int t = threadIdx.x; t < nu * PA_ROWS; t += 256) {
        const int r = t / nu, u = t - r * nu, py = py0 + r;
        if (py >= th) break;
        int sy = py - J.pad_y;
        sy = sy < 0 ? 0 : sy > J.dh - 1 ? J.dh - 1 : sy;
        const uint8_t PA_GLOBAL *srow = PA_AS_GLOBAL(const uint8_t, J.src + (size_t)(sy * J.step) * J.sstride);
        uint8_t PA_GLOBAL       *drow = PA_AS_GLOBAL(uint8_t, J.dst + (size_t)py * J.dstride + 16 * u);
        const int      px0 = 16 * u - J.pad_x;
        uint32_t       w[4];
        if (px0 >= 0 && px0 + 15 < J.dw) { /* interior */
            if (J.step == 1) {
                const pa_u32x4 d = *(const pa_u32x4 PA_GLOBAL *)(srow + px0);
                w[0] = d.x; w[1] = d.y; w[2] = d.z; w[3] = d.w;
            } else {
                _Pragma("unroll") for (int k = 0; k < 4; k++) w[k] = pa_dword_inside(srow + (px0 + 4 * k) * J.step, J.step);
            }
        } else { /* the replicated borders */
            _Pragma("unroll") for (int k = 0; k < 4; k++) {
                w[k] = 0;
                _Pragma("unroll") for (int b = 0; b < 4; b++) {
                    int sx = px0 + 4 * k + b;
                    sx = sx < 0 ? 0 : sx > J.dw - 1 ? J.dw - 1 : sx;
                    w[k] |= (uint32_t)srow[sx * J.step] << (8 * b);
                }
            }
        }
        if (16 * u + 15 < tw) {
            pa_u32x4 o; o.x = w[0]; o.y = w[1]; o.z = w[2]; o.w = w[3];
            *(pa_u32x4 PA_GLOBAL *)drow = o;
        } else
            for (int b = 0; b < 16 && 16 * u + b < tw; b++) drow[b] = (uint8_t)(w[b >> 2] >> (8 * (b & 3)));
    }
}
} // namespace

namespace {
/* one wave per SB: lane b = 8x8 block b (raster); 16x16 / 32x32 / 64x64 by the first 16 / 4 / 1 lanes through LDS */
__global__ __launch_bounds__(256) void svt_pa_meanvar_kernel(svt_plane pl, int nx, int n_sb, uint8_t *__restrict__ mean_out,
                                                             uint16_t *__restrict__ var_out) {
    __shared__ uint64_t s_m[4][85], s_q[4][85]; /* mean << 8 and mean of squares << 16, per wave */
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, sb = blockIdx.x * 4 + w;
    if (sb >= n_sb) return;
    const uint8_t PA_GLOBAL *p = PA_AS_GLOBAL(const uint8_t, pl.buf) + (size_t)(pl.origin_y + (sb / nx) * 64 + (lane >> 3) * 8) * pl.stride + pl.origin_x + (sb % nx) * 64 + (lane & 7) * 8;
    uint32_t sum = 0, sq = 0;
    _Pragma("unroll") for (int r = 0; r < 8; r += 2) { /* rows 0, 2, 4, 6 */
        const uint8_t PA_GLOBAL *row = p + (size_t)r * pl.stride;
        _Pragma("unroll") for (int x = 0; x < 8; x++) { const uint32_t v = row[x]; sum += v; sq += v * v; }
    }
    uint64_t *m = s_m[w], *q = s_q[w];
    m[21 + lane] = (uint64_t)sum << 3;
    q[21 + lane] = (uint64_t)sq << 11;
    /* LDS accesses of one wave are ordered: the levels can follow each other without a workgroup barrier */
    if (lane < 16) {
        const int b = 21 + (lane >> 2) * 16 + (lane & 3) * 2;
        m[5 + lane] = (m[b] + m[b + 1] + m[b + 8] + m[b + 9]) >> 2;
        q[5 + lane] = (q[b] + q[b + 1] + q[b + 8] + q[b + 9]) >> 2;
    }
    if (lane < 4) {
        const int b = 5 + (lane >> 1) * 8 + (lane & 1) * 2;
        m[1 + lane] = (m[b] + m[b + 1] + m[b + 4] + m[b + 5]) >> 2;
        q[1 + lane] = (q[b] + q[b + 1] + q[b + 4] + q[b + 5]) >> 2;
    }
    if (lane == 0) {
        m[0] = (m[1] + m[2] + m[3] + m[4]) >> 2;
        q[0] = (q[1] + q[2] + q[3] + q[4]) >> 2;
    }
    for (int i = lane; i < 85; i += 64) {
        mean_out[(size_t)sb * 85 + i] = (uint8_t)(m[i] >> 8);
        var_out[(size_t)sb * 85 + i]  = (uint16_t)((q[i] - m[i] * m[i]) >> 16);
    }
}
} // namespace

extern "C" int32_t svt_hip_pa_mean_variance_device(svt_hip_ctx *ctx, const svt_plane *full, uint8_t *d_mean, uint16_t *d_var) {
    if (!ctx || !full || !full->buf || !d_mean || !d_var) return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "pa: null argument");
    const int nx = (full->width + 63) / 64, ny = (full->height + 63) / 64, n_sb = nx * ny;
    if (full->stride < full->width + 2 * full->origin_x) return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "pa: plane geometry");
    if (nx * 64 - full->width > full->origin_x || ny * 64 - full->height > full->origin_y)
        return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "pa: padding smaller than the partial SB");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipEventRecord(ctx->ev_start, ctx->stream));
    hipLaunchKernelGGL(svt_pa_meanvar_kernel, dim3((n_sb + 3) / 4), dim3(256), 0, ctx->stream, *full, nx, n_sb, d_mean, d_var);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(ctx->ev_stop, ctx->stream));
    ctx->timed = 1;
    return SVT_HIP_OK;
}

extern "C" int32_t svt_hip_pa_prepare_batch_device(svt_hip_ctx *ctx, int32_t n_pics, const uint8_t *const *d_luma,
                                                   const int32_t *luma_stride, const svt_pa_picture *out, int32_t make_quarter) {
    if (!ctx || n_pics < 1 || !d_luma || !luma_stride || !out) return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "pa: null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    const int per = make_quarter ? 3 : 2, n_jobs = n_pics * per;
    pa_job *h = nullptr, *d = nullptr;
    if (svt_ctx_stage(ctx, sizeof(pa_job) * (size_t)n_jobs, (void **)&h, (void **)&d)) return svt_set_error(SVT_HIP_ERR_NO_RESOURCES, "pa: scratch");
    int rows = 0, k = 0;
    for (int i = 0; i < n_pics; i++) {
        const svt_plane *pl[3] = {&out[i].full, &out[i].quarter, &out[i].sixteenth};
        const int        W = out[i].full.width, H = out[i].full.height;
        if (!d_luma[i] || W < 8 || H < 8 || (W & 7) || (H & 7)) return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "pa: width/height must be multiples of 8");
        for (int s = 0; s < 3; s++) {
            if (s == 1 && !make_quarter) continue;
            const int step = 1 << s;
            if (!pl[s]->buf || pl[s]->width != W / step || pl[s]->height != H / step || pl[s]->stride < W / step + 2 * pl[s]->origin_x)
                return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "pa: destination plane geometry");
            pa_job &J = h[k++];
            J.src = d_luma[i]; J.dst = (uint8_t *)pl[s]->buf; J.sstride = luma_stride[i]; J.dstride = pl[s]->stride; J.step = step;
            J.pad_x = pl[s]->origin_x; J.pad_y = pl[s]->origin_y; J.dw = W / step; J.dh = H / step; J.row0 = rows;
            rows += (J.dh + 2 * J.pad_y + PA_ROWS - 1) / PA_ROWS;
        }
    }
    HIP_TRY(hipMemcpyAsync(d, h, sizeof(pa_job) * (size_t)n_jobs, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipEventRecord(ctx->ev_start, ctx->stream));
    hipLaunchKernelGGL(svt_pa_plane_kernel, dim3(rows), dim3(256), 0, ctx->stream, (const pa_job *)d, n_jobs);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(ctx->ev_stop, ctx->stream));
    svt_ctx_stage_commit(ctx);
    ctx->timed = 1;
    return SVT_HIP_OK;
}
