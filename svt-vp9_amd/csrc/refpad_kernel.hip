/*
 * refpad_kernel.hip -- deblocked reconstruction -> reference-picture form, in place, on gfx950.
 *
 * Replaces pad_ref_and_set_flags (Source/Lib/Codec/EbEncDecProcess.c:4822-4851, called right after the loop filter at :5696)
 * -> eb_vp9_generate_padding (Source/Lib/Codec/EbMcp.c:17-58) on Y with (origin_x, origin_y) and on Cb / Cr with half of it.
 * In the reference the reconstruction buffer of a reference picture IS its reference picture (recon refs are allocated with
 * 64 + 16 = 80 samples of border, Codec/EbEncHandle.c:968-971); the padding replicates the picture's edge samples into that
 * border -- horizontally first, then the padded first / last row upwards / downwards -- so every border sample equals
 * pic[clamp(y)][clamp(x)].  This kernel writes exactly the border samples with that formula (the interior is not touched, so
 * there is no ordering between reads and writes): the traffic is the border's 1.6 MB per 4K picture, not the picture's 12.4.
 * It is the step that closes the loop of the path: svt_lf_kernel's output becomes what svt_mc_kernel reads for the pictures of
 * the next temporal layer (and what svt_hip_ref_handoff_device ships in split-GOP mode).
 */
#include <hip/hip_runtime.h>
#include "svt_ctx.h"

#define RP_GLOBAL __attribute__((address_space(1)))
#define RP_AS_GLOBAL(T, p) ((T RP_GLOBAL *)(uintptr_t)(p))

namespace {
struct rp_job {
    uint8_t *pic;    /* plane sample (0,0) */
    int32_t  stride, w, h, pad_x, pad_y;
    int32_t  wg0;    /* first workgroup of this job in the flattened grid */
};

constexpr int RP_ROWS = 8; /* padded rows per workgroup */

typedef uint32_t rp_u32x4 __attribute__((ext_vector_type(4), aligned(4)));

/* One workgroup = RP_ROWS consecutive rows of the padded plane.  Rows above / below the picture are written whole
 * (w + 2 pad_x samples: the clamped picture row with its two replicated ends), rows of the picture only their two ends.
 * A thread writes 16 bytes at a time where the row piece allows it. */
__global__ __launch_bounds__(256) void svt_refpad_kernel(const rp_job *__restrict__ jobs, int n_jobs) {
    int j = 0;
    while (j + 1 < n_jobs && (int)blockIdx.x >= jobs[j + 1].wg0) j++;
    const rp_job J = jobs[j];
    const int    py0 = ((int)blockIdx.x - J.wg0) * RP_ROWS - J.pad_y;
    /* dword path: every row piece starts on a 4-byte boundary and no dword straddles picture / border */
    const bool   vec = !((J.stride | J.pad_x | J.w) & 3) && !((uintptr_t)J.pic & 3);
    for (int r = 0; r < RP_ROWS; r++) {
        const int py = py0 + r;
        if (py >= J.h + J.pad_y) break;
        const int sy = py < 0 ? 0 : py > J.h - 1 ? J.h - 1 : py;
        const uint8_t RP_GLOBAL *srow = RP_AS_GLOBAL(const uint8_t, J.pic + (ptrdiff_t)sy * J.stride);
        uint8_t RP_GLOBAL       *drow = RP_AS_GLOBAL(uint8_t, J.pic + (ptrdiff_t)py * J.stride);
        const uint32_t lft = srow[0] * 0x01010101u, rgt = srow[J.w - 1] * 0x01010101u;
        if (py >= 0 && py < J.h) { /* a picture row: the two ends only */
            if (vec) {
                const int nd = J.pad_x >> 2; /* dwords per end */
                for (int t = threadIdx.x; t < 2 * nd; t += 256) {
                    const bool right = t >= nd;
                    *(uint32_t RP_GLOBAL *)(drow + (right ? J.w + 4 * (t - nd) : 4 * t - J.pad_x)) = right ? rgt : lft;
                }
            } else {
                for (int t = threadIdx.x; t < 2 * J.pad_x; t += 256) {
                    const bool right = t >= J.pad_x;
                    drow[right ? J.w + (t - J.pad_x) : t - J.pad_x] = (uint8_t)(right ? rgt : lft);
                }
            }
        } else if (vec) { /* a border row, in 16-byte pieces of the padded row (its first byte is 4-byte aligned) */
            const int tw = J.w + 2 * J.pad_x, nu = (tw + 15) >> 4;
            for (int u = threadIdx.x; u < nu; u += 256) {
                const int px0 = 16 * u - J.pad_x; /* picture column of the piece's first byte */
                uint32_t  d[4];
                _Pragma("unroll") for (int k = 0; k < 4; k++) {
                    const int x = px0 + 4 * k; /* pad_x and w are multiples of 4: a dword is never split between regions */
                    d[k] = x < 0 ? lft : x >= J.w ? rgt : *(const uint32_t RP_GLOBAL *)(srow + x);
                }
                if (16 * u + 16 <= tw) {
                    rp_u32x4 o; o.x = d[0]; o.y = d[1]; o.z = d[2]; o.w = d[3];
                    *(rp_u32x4 RP_GLOBAL *)(drow + px0) = o;
                } else
                    for (int k = 0; 16 * u + 4 * k < tw; k++) *(uint32_t RP_GLOBAL *)(drow + px0 + 4 * k) = d[k];
            }
        } else {
            for (int t = threadIdx.x; t < J.w + 2 * J.pad_x; t += 256) {
                int sx = t - J.pad_x;
                sx = sx < 0 ? 0 : sx > J.w - 1 ? J.w - 1 : sx;
                drow[t - J.pad_x] = srow[sx];
            }
        }
    }
}
} // namespace

extern "C" int32_t svt_hip_ref_pad_batch_device(svt_hip_ctx *ctx, int32_t n_pics, const svt_yuv_planes *pics, int32_t pad_x, int32_t pad_y) {
    if (!ctx || n_pics < 1 || !pics) return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "ref_pad: null argument");
    if (pad_x < 0 || pad_y < 0 || (pad_x & 1) || (pad_y & 1)) return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "ref_pad: padding must be even and >= 0");
    if (!pad_x && !pad_y) return SVT_HIP_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    const int n_jobs = 3 * n_pics;
    rp_job   *h = nullptr, *d = nullptr;
    if (svt_ctx_stage(ctx, sizeof(rp_job) * (size_t)n_jobs, (void **)&h, (void **)&d)) return svt_set_error(SVT_HIP_ERR_NO_RESOURCES, "ref_pad: scratch");
    int wgs = 0, k = 0;
    for (int i = 0; i < n_pics; i++) {
        const svt_yuv_planes &P = pics[i];
        if (!P.y || !P.u || !P.v || P.width < 2 || P.height < 2 || (P.width & 1) || (P.height & 1))
            return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "ref_pad: picture geometry");
        if (P.y_stride < P.width + 2 * pad_x || P.uv_stride < P.width / 2 + pad_x)
            return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "ref_pad: stride smaller than the padded row");
        uint8_t *pl[3] = {P.y, P.u, P.v};
        for (int c = 0; c < 3; c++) {
            rp_job &J = h[k++];
            const int sh = c ? 1 : 0;
            J.pic = pl[c]; J.stride = c ? P.uv_stride : P.y_stride; J.w = P.width >> sh; J.h = P.height >> sh;
            J.pad_x = pad_x >> sh; J.pad_y = pad_y >> sh; J.wg0 = wgs;
            wgs += (J.h + 2 * J.pad_y + RP_ROWS - 1) / RP_ROWS;
        }
    }
    HIP_TRY(hipMemcpyAsync(d, h, sizeof(rp_job) * (size_t)n_jobs, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipEventRecord(ctx->ev_start, ctx->stream));
    hipLaunchKernelGGL(svt_refpad_kernel, dim3(wgs), dim3(256), 0, ctx->stream, (const rp_job *)d, n_jobs);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(ctx->ev_stop, ctx->stream));
    svt_ctx_stage_commit(ctx);
    ctx->timed = 1;
    return SVT_HIP_OK;
}
