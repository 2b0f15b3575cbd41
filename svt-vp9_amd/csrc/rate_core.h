/* rate_core.h -- the per-position arithmetic of coeff_rate_estimate (Codec/EbRateDistortionCost.c:55-172 = libvpx cost_coeffs,
 * use_fast_coef_costing = 0 branch :131-169), shared by the stand-alone rate kernel (rate_kernel.hip) and the rate pass fused
 * behind the quantiser (tq_kernel.hip). */
#ifndef SVT_RATE_CORE_H
#define SVT_RATE_CORE_H
#include <hip/hip_runtime.h>
#include "../../include/svtvp9_hip.h"

namespace {

constexpr int RATE_SLICE = 6 * 2 * 6 * 12;           /* dwords of one token_costs[tx_size][plane_type][is_inter] slice */
/* element offset of the {scan[n], neighbors[2 (n + 1)]} table of (tx_size, tx_type) in the canonical scan array: tables of
 * tx_type 0..3 of 4x4, then 8x8, 16x16, 32x32 (all four slots present for every size, 3 n + 2 entries each) */
__host__ __device__ __forceinline__ int rate_scan_offset(int tx_size, int tx_type) {
    int off = 0;
    for (int s = 0; s < tx_size; s++) off += 4 * (3 * (16 << (2 * s)) + 2);
    return off + tx_type * (3 * (16 << (2 * tx_size)) + 2);
}

__device__ __forceinline__ int token_of(int v) { /* VPX/vp9_tokenize.c:36-50, VPX/vp9_entropy.h:28-52 */
    /* 0..4 -> the value; 5-6, 7-10, 11-18, 19-34, 35-66 -> CAT1..CAT5 (5..9), >= 67 -> CAT6 (10): the categories are the
     * octaves of |v| - 3 */
    const int a = v < 0 ? -v : v;
    const int oct = 35 - __builtin_clz((unsigned)(a - 3) | 1u); /* 4 + floor(log2(a - 3)) for a >= 5 */
    return a < 5 ? a : (oct < 10 ? oct : 10);
}
/* eb_vp9_pt_energy_class[token] = {0,1,2,3,3,4,4,5,5,5,5,5}: straight from |v| (0, 1, 2, 3-4, 5-10, >= 11) */
__device__ __forceinline__ int energy_of_value(int v) {
    int a = v < 0 ? -v : v;
    a = a < 11 ? a : 11;
    return (int)((0x544444433210ull >> (4 * a)) & 0xf);
}
__device__ __forceinline__ int band_of(int c, int tx4x4) { return c == 0 ? 0 : c < 3 ? 1 : c < 6 ? 2 : c < 10 ? 3 : c < (tx4x4 ? 13 : 21) ? 4 : 5; }

/* bits of the scan positions lane, lane + STRIDE, .. (< n, <= eob) of one block; q = the block's coefficients, scan / nb / tc =
 * its scan order and cost slice (each either in LDS or in global memory) */
template <int STRIDE>
__device__ __forceinline__ int rate_positions(const int16_t *q, const int16_t *scan, const int16_t *nb, const uint32_t *tc, const svt_rate_tables *T,
                                              int lane, int eob, int n, int ts, int ctx0) {
    int sum = 0;
    _Pragma("unroll 2") for (int c = lane; c <= eob && c < n; c += STRIDE) {
        /* the three table reads of a position are independent of each other: issue them before the coefficient reads */
        const uint32_t nn = c ? *(const uint32_t *)(nb + 2 * c) : 0u;
        const int      rc = c ? scan[c] : 0, rp = c ? scan[c - 1] : 0;
        int            pt = ctx0, pz = 0, band = 0;
        if (c) {
            pt   = (1 + energy_of_value(q[(int16_t)(nn & 0xffff)]) + energy_of_value(q[(int16_t)(nn >> 16)])) >> 1;
            pz   = q[rp] == 0;
            band = band_of(c, ts == 0);
        }
        if (c == eob) { /* EOB token (the block is not full) */
            sum += (int)tc[((band * 2 + 0) * 6 + pt) * 12 + 11];
        } else {
            const int v = q[rc], tok = token_of(v);
            int       cost;
            if (tok == 10) { /* vp9_get_token_cost, VPX/vp9_tokenize.h:118-127 */
                const int extra = (v < 0 ? -v : v) - 67;
                cost = T->cat6_low_cost[extra & 0xff] + T->cat6_high_cost[extra >> 8];
            } else {
                cost = T->value_cost[v + 66];
            }
            sum += cost + (int)tc[((band * 2 + pz) * 6 + pt) * 12 + tok];
        }
    }
    return sum;
}


} // namespace
#endif
