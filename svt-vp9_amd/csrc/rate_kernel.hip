/*
 * rate_kernel.hip -- coefficient rate estimation of batches of quantised transform blocks (gfx950).
 *
 * Replaces coeff_rate_estimate (Source/Lib/Codec/EbRateDistortionCost.c:55-172 = libvpx cost_coeffs, the
 * use_fast_coef_costing = 0 branch :131-169) as perform_dist_rate_calc calls it for every transform block
 * (Codec/EbEncDecProcess.c:734-745).
 *
 * The reference walks the scan serially, but nothing in the walk is a true chain: the token of a position depends on
 * its own coefficient only, its context on the energy classes of two earlier-scanned neighbours (= their
 * coefficients), the "previous token was ZERO" switch on the previous position's coefficient, the band on the
 * position.  So 16 lanes take one block (most blocks are 4x4 / 8x8 with a short scan; a wave holds four blocks), lane l
 * the scan positions l, l + 16, ..; position eob (if the block is not full) contributes the EOB token; a reduction
 * over the 16 lanes gives the block's bits.  The block's coefficients are first copied to LDS with 16-byte loads (the
 * three coefficient gathers per position then cost an LDS access instead of a global one); table reads are gathers
 * from the 55 KB cost table (L1/L2 resident).
 */
#include <hip/hip_runtime.h>
#include "svt_ctx.h"

#include "rate_core.h"

namespace {

constexpr int RATE_SCAN4 = 16 + 2 * 17, RATE_SCAN8 = 64 + 2 * 65; /* int16 entries of a 4x4 / 8x8 {scan, neighbors} table */
constexpr size_t RATE_SCAN_PREFETCH = 4 * RATE_SCAN4 + 4 * RATE_SCAN8; /* = SVT_RATE_SCAN_MIN_ENTRIES (976) */
static_assert(RATE_SCAN_PREFETCH == SVT_RATE_SCAN_MIN_ENTRIES, "header constant");

/* Persistent workgroups: a workgroup first copies what the small blocks need -- the eight cost slices of 4x4 / 8x8 blocks
 * (27 KB), their eight scan orders (2 KB, when the caller's scan array has the canonical layout of
 * [tx_size][tx_type] tables) -- into LDS once, then walks chunks of 16 blocks.  For a 4x4 / 8x8 block the only global
 * reads left are its descriptor and its coefficients. */
__global__ __launch_bounds__(256) void svt_rate_kernel(const int16_t *__restrict__ qcoeff, const svt_rate_block *__restrict__ blocks, int n_blocks,
                                                       const svt_rate_tables *__restrict__ T, const int16_t *__restrict__ scan_all,
                                                       int32_t *__restrict__ bits) {
    __shared__ uint32_t s_tc[8 * RATE_SLICE];
    __shared__ int16_t  s_scan[4 * RATE_SCAN4 + 4 * RATE_SCAN8];
    __shared__ uint4    s_q[16][8];
    {
        const uint32_t *g = &T->token_costs[0][0][0][0][0][0][0]; /* slices of tx_size 0 and 1 are the first 8 */
        for (int i = threadIdx.x; i < 8 * RATE_SLICE; i += 256) s_tc[i] = g[i];
        const uint32_t *gs = (const uint32_t *)scan_all;          /* canonical layout: the 4x4 tables first, then the 8x8 ones */
        for (int i = threadIdx.x; i < (4 * RATE_SCAN4 + 4 * RATE_SCAN8) / 2; i += 256) ((uint32_t *)s_scan)[i] = gs[i];
    }
    __syncthreads();
    const int grp = threadIdx.x >> 4, lane = threadIdx.x & 15;
    for (int base = blockIdx.x * 16; base < n_blocks; base += gridDim.x * 16) {
        const int b = base + grp;
        if (b < n_blocks) { /* no barrier inside: the 16 lanes of a group sit in one wave, whose LDS accesses are ordered */
            const uint4     kw = *(const uint4 *)(blocks + b);
            const uint32_t  coeff_off = kw.x, scan_off = kw.y;
            const int       eob = (int)(kw.z & 0xffff), ts = (int)((kw.z >> 16) & 0xff), ptype = (int)(kw.z >> 24);
            const int       inter = (int)(kw.w & 0xff), ctx0 = (int)((kw.w >> 8) & 0xff);
            const int       n = 16 << (2 * ts);
            const int       slice = (ts * 2 + ptype) * 2 + inter;
            int             sum;
            if (ts <= 1) {
                /* canonical scan offsets: 4x4 table tt at tt * 50, 8x8 table tt at 200 + tt * 194; anything else is read in place */
                const int       tt = ts == 0 ? (int)scan_off / RATE_SCAN4 : ((int)scan_off - 4 * RATE_SCAN4) / RATE_SCAN8;
                const bool      canon = ts == 0 ? (scan_off == (uint32_t)(tt * RATE_SCAN4) && tt < 4)
                                                : (scan_off == (uint32_t)(4 * RATE_SCAN4 + tt * RATE_SCAN8) && tt >= 0 && tt < 4);
                const int16_t  *sc_l = s_scan + scan_off, *sc_g = scan_all + scan_off;
                if (eob && lane < (n >> 3)) s_q[grp][lane] = ((const uint4 *)(qcoeff + coeff_off))[lane];
                if (canon) sum = rate_positions<16>((const int16_t *)s_q[grp], sc_l, sc_l + n, s_tc + slice * RATE_SLICE, T, lane, eob, n, ts, ctx0);
                else sum = rate_positions<16>((const int16_t *)s_q[grp], sc_g, sc_g + n, s_tc + slice * RATE_SLICE, T, lane, eob, n, ts, ctx0);
            } else {
                const int16_t *sc_g = scan_all + scan_off;
                sum = rate_positions<16>(qcoeff + coeff_off, sc_g, sc_g + n, &T->token_costs[0][0][0][0][0][0][0] + slice * RATE_SLICE, T, lane, eob, n, ts, ctx0);
            }
            _Pragma("unroll") for (int off = 8; off; off >>= 1) sum += __shfl_xor(sum, off);
            if (lane == 0) bits[b] = sum;
        }
    }
}
} // namespace

extern "C" int32_t svt_hip_coeff_rate_batch_device(svt_hip_ctx *ctx, const int16_t *d_qcoeff, const svt_rate_block *d_blocks, int32_t n_blocks,
                                                   const svt_rate_tables *d_tables, const int16_t *d_scan, int32_t *d_bits) {
    if (!ctx || !d_qcoeff || !d_blocks || n_blocks < 1 || !d_tables || !d_scan || !d_bits)
        return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "rate: null argument");
    if (((uintptr_t)d_scan & 3) || ((uintptr_t)d_blocks & 15) || ((uintptr_t)d_qcoeff & 15))
        return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "rate: scan / block / coefficient arrays must be 4 / 16 / 16-byte aligned");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipEventRecord(ctx->ev_start, ctx->stream));
    const int dev_cus = ctx->cu_count;
    const int want = (n_blocks + 15) / 16, cap = dev_cus * 5; /* 31 KB of LDS per workgroup: five per CU */
    hipLaunchKernelGGL(svt_rate_kernel, dim3(want < cap ? want : cap), dim3(256), 0, ctx->stream, d_qcoeff, d_blocks, n_blocks, d_tables, d_scan, d_bits);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(ctx->ev_stop, ctx->stream));
    ctx->timed = 1;
    return SVT_HIP_OK;
}

extern "C" int32_t svt_hip_coeff_rate_batch(svt_hip_ctx *ctx, const int16_t *qcoeff, size_t coeff_count, const svt_rate_block *blocks, int32_t n_blocks,
                                            const svt_rate_tables *tables, const int16_t *scan, size_t scan_count, int32_t *bits) {
    if (!ctx || !qcoeff || !blocks || n_blocks < 1 || !tables || !scan || !bits) return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "rate: null argument");
    for (int i = 0; i < n_blocks; i++) {
        const size_t n = (size_t)16 << (2 * blocks[i].tx_size);
        if (blocks[i].tx_size > 3 || blocks[i].plane_type > 1 || blocks[i].is_inter > 1 || blocks[i].ctx > 2 || blocks[i].eob > n ||
            blocks[i].coeff_off + n > coeff_count || (blocks[i].coeff_off & 7) || blocks[i].scan_off + 3 * n + 2 > scan_count || (blocks[i].scan_off & 1))
            return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "rate: bad block");
    }
    HIP_TRY(hipSetDevice(ctx->device));
    int16_t          *dq = (int16_t *)svt_ctx_slot(ctx, 34, sizeof(int16_t) * coeff_count);
    svt_rate_block   *db = (svt_rate_block *)svt_ctx_slot(ctx, 35, sizeof(svt_rate_block) * (size_t)n_blocks);
    svt_rate_tables  *dt = (svt_rate_tables *)svt_ctx_slot(ctx, 36, sizeof(svt_rate_tables));
    /* the kernel's prologue copies the first RATE_SCAN_PREFETCH entries whatever the blocks use (see the header) */
    const size_t      scan_alloc = scan_count > RATE_SCAN_PREFETCH ? scan_count : RATE_SCAN_PREFETCH;
    int16_t          *ds = (int16_t *)svt_ctx_slot(ctx, 37, sizeof(int16_t) * scan_alloc);
    int32_t          *dbits = (int32_t *)svt_ctx_slot(ctx, 38, sizeof(int32_t) * (size_t)n_blocks);
    if (!dq || !db || !dt || !ds || !dbits) return svt_set_error(SVT_HIP_ERR_NO_RESOURCES, "rate: device buffers");
    HIP_TRY(hipMemcpyAsync(dq, qcoeff, sizeof(int16_t) * coeff_count, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(db, blocks, sizeof(svt_rate_block) * (size_t)n_blocks, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(dt, tables, sizeof(svt_rate_tables), hipMemcpyHostToDevice, ctx->stream));
    if (scan_alloc > scan_count) HIP_TRY(hipMemsetAsync(ds + scan_count, 0, sizeof(int16_t) * (scan_alloc - scan_count), ctx->stream));
    HIP_TRY(hipMemcpyAsync(ds, scan, sizeof(int16_t) * scan_count, hipMemcpyHostToDevice, ctx->stream));
    int32_t rc = svt_hip_coeff_rate_batch_device(ctx, dq, db, n_blocks, dt, ds, dbits);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(bits, dbits, sizeof(int32_t) * (size_t)n_blocks, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return SVT_HIP_OK;
}
