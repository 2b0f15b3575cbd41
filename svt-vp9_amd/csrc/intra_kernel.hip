/*
 * intra_kernel.hip -- the encode pass of an INTRA picture (gfx950): reference samples, the ten VP9 predictors, transform /
 * quantisation / reconstruction of every block, in the dependency order intra prediction imposes.
 *
 * Replaces, for a picture whose blocks are all intra (key frames and intra-refresh pictures), the per-block body of encode_pass_sb
 * (Source/Lib/Codec/EbEncDecProcess.c:3680-4160):
 *   generate_intra_reference_samples (:1128-1310; the neighbour arrays of Codec/EbNeighborArrays.c:107-240 hold the block's
 *   unfiltered reconstruction, so the reference samples are read straight from the reconstruction plane here)
 *   -> intra_prediction (Codec/EbIntraPrediction.c:16) -> eb_vp9_predict_intra_block / build_intra_predictors
 *      (VPX/vp9_reconintra.c:249-408) -> the predictors of VPX/intrapred.c:22-416
 *   -> perform_coding_loop (:365-587) with the transform type of the luma mode (eb_vp9_intra_mode_to_tx_type_lookup,
 *      vp9_reconintra.c:20-31; chroma and 32x32: DCT_DCT) -> tq_block_body (tq_core.h, the body the batch kernels use).
 * Scope: blocks of 4x4 (four to an 8x8 unit, each with its own mode), 8x8, 16x16 and 32x32 with the transform of their own size, inside
 * the picture -- what the reference codes in an intra picture whose dimensions are multiples of 8.  None of them reads the
 * above-right NEIGHBOUR BLOCK (have_right = 0 for blocks >= 8x8, EbEncDecProcess.c:1146; a 4x4 block in the left half of its unit
 * reads four above-right samples, which belong to the unit above or to the 4x4 block coded just before it) and none crosses the
 * picture edge.
 *
 * Parallelism: a block needs the reconstruction of its left, above and above-left neighbours -- and nothing else: without the
 * above-right neighbour the reference's z-order is only ONE of the orders that give its result.  The picture is cut into 32x32 luma
 * areas (the largest block here); the blocks of an area are coded one after the other (z-order) and area (r, c) can start when
 * (r, c - 1) and (r - 1, c) are done: an anti-diagonal wavefront over the areas, times three independent planes.  One 64-lane
 * workgroup codes one (area, plane) at a time; the workgroups (one per CU, persistent) take TICKETS (an atomic counter) that enumerate
 * the (area, plane) pairs diagonal by diagonal, so a workgroup only ever waits for tickets smaller than its own -- which are held by
 * workgroups that already run or are done: no deadlock whatever the dispatch order, no co-residency requirement.  A block is N x N
 * with N lanes active (lane i = row i of the prediction, then column / row i of the transform); the chain of dependent blocks, not
 * the lane count, bounds the speed: a 2160p key frame is 187 diagonals of at most 68 areas.  This kernel is latency-bound by design (one picture in a GOP); it shares the GPU with
 * the batches of the inter pictures running beside it.
 * Visibility between workgroups (other CUs, other XCDs' L2): release fence + flag store when an area is done, flag load + acquire
 * fence before the first reference-sample load; inside a workgroup the block's stores are drained (workgroup fence) before the
 * next block reads them.
 */
#include <hip/hip_runtime.h>
#include "tq_core.h"
#include "encdec_core.h"
#include <stdlib.h>

namespace {

struct intra_pic_dev {
    const uint8_t *src[3];
    uint8_t       *pred[3];  /* may be null: the prediction is then not stored */
    uint8_t       *rec[3];
    int32_t        src_stride[2], pred_stride[2], rec_stride[2];
    const svt_lf_mode_info *mi;
    int32_t        mi_stride, mi_rows, mi_cols, sb_cols, sb_rows, width, height;
    const svt_quant_tables *qtabs;
    const int16_t *iscan;
    uint32_t       iscan_off[16];
    int16_t       *qcoeff, *dqcoeff;
    uint16_t      *eob_map;
    uint8_t       *nz;
    int32_t       *sync;    /* [0] ticket counter, [2 + plane * n_cell + cell] done flags (16x16 luma cells); zeroed before the launch */
    int32_t       *status;  /* |= 1: a malformed grid was seen */
    int32_t        mixed;   /* 1: an inter picture with some intra blocks: inter blocks are skipped (the batch coded them) */
};

/* eb_vp9_intra_mode_to_tx_type_lookup (VPX/vp9_reconintra.c:20-31): DC, V, H, D45, D135, D117, D153, D207, D63, TM */
__device__ __forceinline__ int intra_tx_type(int mode) { return (int)((0x3122130210ull >> (4 * mode)) & 3); }

#define AVG2(a, b) (((a) + (b) + 1) >> 1)
#define AVG3(a, b, c) (((a) + 2 * (b) + (c) + 2) >> 2)

/* Row r of the N x N prediction, packed.  e[k + 32] = B(k): B(0) the corner sample above[-1], B(k > 0) = above[k - 1] (2N samples:
 * the second N replicate above[N - 1], or are the true above-right samples), B(-k) = left[k - 1].  Closed forms of the procedures of
 * VPX/intrapred.c (d207 :22, d63 :43, d45 :59, d117 :75, d135 :99, d153 :120, v / h / tm :140-172, the four DC forms :174-236,
 * the 4x4 forms of d45 / d63 :349-416 which read the above-right samples). */
template <int N> __device__ __forceinline__ void intra_pred_row(const uint8_t *e, int mode, int r, int have_left, int have_top, uint32_t (&prow)[N / 4]) {
    auto B = [&](int k) -> int { return (int)e[k + 32]; };
    int dc = 128;
    if (mode == 0 && (have_left || have_top)) {
        int sum = 0, cnt = 0;
        if (have_top) { _Pragma("unroll") for (int j = 0; j < N; j++) sum += B(j + 1); cnt += N; }
        if (have_left) { _Pragma("unroll") for (int j = 0; j < N; j++) sum += B(-(j + 1)); cnt += N; }
        dc = (sum + (cnt >> 1)) / cnt;
    }
    _Pragma("unroll") for (int q = 0; q < N / 4; q++) prow[q] = 0;
    /* the mode is uniform over the wave: one branch, then N samples without control flow */
    auto fill = [&](auto f) {
        _Pragma("unroll") for (int c = 0; c < N; c++) prow[c >> 2] |= (uint32_t)(f(c) & 0xff) << (8 * (c & 3));
    };
    switch (mode) {
    case 0: fill([&](int) { return dc; }); break;
    case 1: fill([&](int c) { return B(c + 1); }); break;
    case 2: fill([&](int) { return B(-(r + 1)); }); break;
    case 3: /* D45 */
        fill([&](int c) {
            if (N == 4) return (r + c == 6) ? B(8) : AVG3(B(r + c + 1), B(r + c + 2), B(r + c + 3));
            return (r + c < N - 1) ? AVG3(B(r + c + 1), B(r + c + 2), B(r + c + 3)) : B(N);
        });
        break;
    case 4: fill([&](int c) { const int p = c - r; return AVG3(B(p - 1), B(p), B(p + 1)); }); break; /* D135 */
    case 5: /* D117 */
        fill([&](int c) {
            const int h = r >> 1;
            if (c >= h) { const int cc = c - h; return (r & 1) ? AVG3(B(cc - 1), B(cc), B(cc + 1)) : AVG2(B(cc), B(cc + 1)); }
            const int rr = r - 2 * c;
            return AVG3(B(-(rr - 2)), B(-(rr - 1)), B(-rr));
        });
        break;
    case 6: /* D153 */
        fill([&](int c) {
            const int h = c >> 1;
            if (r >= h) { const int rr = r - h; return (c & 1) ? AVG3(B(-rr + 1), B(-rr), B(-rr - 1)) : AVG2(B(-rr), B(-rr - 1)); }
            const int cc = c - 2 * r;
            return AVG3(B(cc - 2), B(cc - 1), B(cc));
        });
        break;
    case 7: /* D207: left[] clamped at N - 1 */
        fill([&](int c) {
            const int idx = r + (c >> 1);
            const int l0 = B(-(min(idx, N - 1) + 1)), l1 = B(-(min(idx + 1, N - 1) + 1)), l2 = B(-(min(idx + 2, N - 1) + 1));
            return (c & 1) ? AVG3(l0, l1, l2) : AVG2(l0, l1);
        });
        break;
    case 8: /* D63 */
        fill([&](int c) {
            const int h = r >> 1;
            if (N > 4 && r >= 2 && c >= N - 1 - h) return B(N);
            return (r & 1) ? AVG3(B(c + h + 1), B(c + h + 2), B(c + h + 3)) : AVG2(B(c + h + 1), B(c + h + 2));
        });
        break;
    default: /* TM */
        fill([&](int c) { const int v = B(-(r + 1)) + B(c + 1) - B(0); return v < 0 ? 0 : v > 255 ? 255 : v; });
        break;
    }
}

/* Visibility of a block's reconstruction to the workgroups that predict from it (other CUs, other XCDs): written through and read
 * with agent-scope accesses (the deblocking kernel's seam-row scheme), NOT with a release / acquire fence pair per area: such a pair
 * is buffer_wbl2 sc1 + buffer_inv sc1 -- write-back and invalidation of the XCD's whole L2 -- 18 000 times per 2160p key frame, paid by
 * every kernel that runs beside this one (measured in the step: a key frame cost + 4 ms of step time with the fences, however few
 * workgroups coded it).  -DINTRA_FENCES restores the fence form (3-12 % faster when the pass has the GPU to itself). */
#ifdef INTRA_FENCES
#define INTRA_WT false
#define INTRA_LD(p) (*(p))
#else
#define INTRA_WT true
#define INTRA_LD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#endif
#ifdef INTRA_PROF
__device__ unsigned long long g_intra_fine[8];
#define IPF(k) do { if (lane == 0) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); atomicAdd(&g_intra_fine[k], n_ - ipf_t); ipf_t = n_; } } while (0)
#else
#define IPF(k) ((void)0)
#endif
/* one N x N transform block of plane `plane` at sample (x0, y0) of that plane; the whole workgroup (one wave) calls it */
template <int N>
__device__ __forceinline__ int intra_block(const intra_pic_dev &P, int plane, int x0, int y0, int mode, int sb, int32_t *tile, uint8_t *edge, int have_right = 0) {
    const int lane = (int)threadIdx.x, c = plane ? 1 : 0;
    const int rs = P.rec_stride[c];
    uint8_t  *rp = P.rec[plane];
    const int have_left = x0 > 0, have_top = y0 > 0;
    const bool active = lane < N;
    const int  i = lane % N;
#ifdef INTRA_PROF
    unsigned long long ipf_t = __builtin_amdgcn_s_memtime();
#endif
    uint32_t   srow[N / 4], prow[N / 4];
    constexpr uintptr_t AM = N >= 16 ? 15 : N - 1;
    {   /* the source row does not depend on the neighbours: its load is issued first and completes under the reference-sample loads */
        const uint8_t *sp = P.src[plane] + (size_t)(y0 + i) * P.src_stride[c] + x0;
        _Pragma("unroll") for (int q = 0; q < N / 4; q++) srow[q] = 0u;
        if (active) row_load<N>(sp, ((uintptr_t)sp & AM) == 0, srow);
    }
    /* reference samples -> LDS (generate_intra_reference_samples, the paths of a block inside the picture) */
    {
        const int j = lane & 31;
        if (lane < 32) { /* left column */
            if (j < N) edge[32 - (j + 1)] = have_left ? INTRA_LD(&rp[(size_t)(y0 + j) * rs + x0 - 1]) : (uint8_t)129;
        } else {         /* above row, replicated to the right */
            if (j < N) {
                const uint8_t a = have_top ? INTRA_LD(&rp[(size_t)(y0 - 1) * rs + x0 + j]) : (uint8_t)127;
                edge[32 + 1 + j] = a;
                /* the right half of the row: true above-right samples for a 4x4 luma block in the left half of its unit (have_right,
                   EbEncDecProcess.c:1146, 1279-1288), copies of the last sample otherwise */
                if (N == 4 && have_right) edge[32 + 1 + N + j] = have_top ? INTRA_LD(&rp[(size_t)(y0 - 1) * rs + x0 + N + j]) : (uint8_t)127;
                else if (j == N - 1) { _Pragma("unroll") for (int q = 0; q < N; q++) edge[32 + 1 + N + q] = a; }
            }
        }
        if (lane == 0) edge[32] = have_top ? (have_left ? INTRA_LD(&rp[(size_t)(y0 - 1) * rs + x0 - 1]) : (uint8_t)129) : (uint8_t)127;
    }
    __syncthreads();
    IPF(0); /* source issue + reference samples into LDS */
    intra_pred_row<N>(edge, mode, i, have_left, have_top, prow);
    {
        if (active && P.pred[plane]) {
            uint8_t *pp = P.pred[plane] + (size_t)(y0 + i) * P.pred_stride[c] + x0;
            row_store<N>(pp, ((uintptr_t)pp & AM) == 0, prow);
        }
    }
    IPF(1); /* prediction (+ its store) */
    svt_tq_block k;
    k.src_off = k.pred_off = 0;
    k.recon_off = (uint32_t)y0 * (uint32_t)rs + (uint32_t)x0;
    k.coeff_off = (uint32_t)sb * SVT_SB_COEFFS + (plane == 0 ? 0u : plane == 1 ? 4096u : 5120u) + svt_zorder4((x0 & (plane ? 31 : 63)) >> 2, (y0 & (plane ? 31 : 63)) >> 2) * 16u;
    k.tx_size = (uint8_t)txcfg<N>::size;
    k.tx_type = (uint8_t)((plane == 0 && N < 32) ? intra_tx_type(mode) : 0);
    k.iscan_off = P.iscan_off[k.tx_size * 4 + k.tx_type];
    k.src_stride = k.pred_stride = 0; k.recon_stride = (uint16_t)rs;
    k.qtab = (uint8_t)c; k.do_recon = 1; k.partial32 = 0; k.pad_[0] = 0;
    const int pw4 = (plane ? P.width >> 1 : P.width) >> 2, w4 = P.width >> 2, h4 = P.height >> 2;
    const int eo = plane == 0 ? 0 : w4 * h4 + (plane == 2 ? (w4 >> 1) * (h4 >> 1) : 0);
    int32_t *t = tile + (lane / N) * (N * (N + 1)); /* the idle slots of the wave run along on tiles of their own */
    const int eob = tq_block_body<N, false, false, INTRA_WT>(k, active, i, t, srow, prow, P.qtabs, P.iscan, P.qcoeff, P.dqcoeff,
                                                   P.eob_map + eo + (y0 >> 2) * pw4 + (x0 >> 2), nullptr, nullptr, nullptr, nullptr, nullptr, rp);
    IPF(2); /* transform / quantisation / reconstruction */
    /* the block's reconstruction is read by the next block of this wave: drain the stores */
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __syncthreads();
    IPF(3); /* store drain */
    return __builtin_amdgcn_readfirstlane(eob);
}

#ifndef INTRA_WAVES_PER_EU
#define INTRA_WAVES_PER_EU 4
#endif
/* Register budget: left alone the compiler takes 308 VGPRs for this kernel -- one such wave per CU, resident for milliseconds, leaves
 * its SIMD room for two motion-estimation waves instead of five, i.e. the CU two ME workgroups instead of five (measured in the step:
 * + 4 ms per key frame, whatever the number of workgroups of this launch).  Held to 128 it spills the 32x32 path's transform rows
 * to scratch, but displaces one ME wave, not three. */
#ifdef INTRA_NUM_VGPR
__attribute__((amdgpu_num_vgpr(INTRA_NUM_VGPR)))
#endif
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(INTRA_WAVES_PER_EU, INTRA_WAVES_PER_EU))) void svt_intra_kernel(const intra_pic_dev P) {
    __shared__ int32_t tile[2 * 32 * 33];
    __shared__ uint8_t edge[128];
    __shared__ int32_t s_ticket;
    /* The unit of scheduling is a 16x16 luma CELL (8x8 in the chroma planes) -- round 6; until then a 32x32 area, whose four 16x16 blocks ran
     * one after the other behind ONE pair of flags: a 2160p key frame was 187 diagonals of 4 block steps.  Intra prediction of blocks >= 8x8
     * never reads the above-right neighbour, so any order in which a block follows its left, above and above-left neighbours gives the
     * reference's result: cells go in anti-diagonal order (374 diagonals of ONE 16x16 step at 2160p), the blocks inside a cell in z-order.
     * A cell's flag says "this cell and everything it depended on is reconstructed and visible".  A 32x32 block covers 2 x 2 cells: it is
     * coded by the ticket of its TOP-RIGHT cell -- every cell it depends on then lies on an earlier diagonal, so a workgroup still only ever
     * waits for smaller tickets -- which sets all four flags; the tickets of its other three cells have nothing to do. */
    const int lane = (int)threadIdx.x, c_cols = (P.width + 15) >> 4, c_rows = (P.height + 15) >> 4, n_cell = c_cols * c_rows;
    const int n_diag = c_rows + c_cols - 1;
    /* one wave per CU on a chain of dependent blocks, beside kernels that fill the SIMDs: it issues rarely, so letting it go first costs
     * the others next to nothing and keeps the chain at the speed it has alone */
    __builtin_amdgcn_s_setprio(3);
  /* workgroups are persistent: each keeps drawing tickets until they run out -- a launch of one workgroup per (cell, plane) would keep
     thousands of them resident, nearly all polling flags of cells many diagonals away */
#ifdef INTRA_PROF
  unsigned long long pf[5] = {0, 0, 0, 0, 0}, pf_n = 0;
#endif
  for (;;) {
    __syncthreads(); /* (s_ticket of the previous round has been read by every lane) */
    if (lane == 0) s_ticket = atomicAdd(&P.sync[0], 1);
    __syncthreads();
#ifdef INTRA_PROF
    if (s_ticket >= 3 * n_cell) { if (lane == 0 && blockIdx.x == 0) printf("[intra-fine] refs %llu predict %llu tq %llu drain %llu (summed over all workgroups so far)\n", g_intra_fine[0], g_intra_fine[1], g_intra_fine[2], g_intra_fine[3]);
      if (lane == 0 && blockIdx.x < 2) printf("[intra-prof] wg %d cells %llu: ticket+map %llu wait %llu work %llu drain %llu publish %llu (100 MHz ticks)\n", (int)blockIdx.x, pf_n, pf[0], pf[1], pf[2], pf[3], pf[4]); break; }
    unsigned long long pt0 = __builtin_amdgcn_s_memtime();
#else
    if (s_ticket >= 3 * n_cell) break;
#endif
    const int ticket = s_ticket, plane = ticket % 3;
    /* the n-th cell in anti-diagonal order (diagonal d = row + col, rows ascending inside a diagonal): cells before diagonal d in closed form
     * (growing part d (d + 1) / 2, then full diagonals of m = min(rows, cols) cells, then the shrinking tail), d by bisection */
    const int n = ticket / 3, m = c_rows < c_cols ? c_rows : c_cols, M = c_rows < c_cols ? c_cols : c_rows;
    int lo = 0, hi = n_diag; /* largest d with count(d) <= n */
    auto count = [&](int d) -> int { /* cells on diagonals 0 .. d-1 */
        const int a = d < m ? d : m;                        /* growing diagonals 0 .. a-1: 1 .. a cells */
        int       t = a * (a + 1) / 2;
        if (d > m) { const int fl = (d < M ? d : M) - m; t += fl * m; }          /* diagonals m .. min(d, M)-1: m cells each */
        if (d > M) { const int k = d - M; t += k * m - k * (k + 1) / 2; }         /* diagonals M .. d-1: m-1, m-2, .. cells */
        return t;
    };
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (count(mid) <= n) lo = mid; else hi = mid; }
    const int d = lo, r_lo = d - (c_cols - 1) > 0 ? d - (c_cols - 1) : 0;
    const int cr = r_lo + (n - count(d)), cc = d - cr;
    const int cell = cr * c_cols + cc;
    int32_t  *done = P.sync + 2 + plane * n_cell;
    /* what the cell belongs to: the unit at the origin of its 2 x 2 cell group decides (a 32x32 block starts there or nowhere in the group), so
       the four cells of a group agree whatever the grid holds */
    const int  br = cr & ~1, bc = cc & ~1;                                      /* first cell of the group */
    const int  ub_r = br * 2, ub_c = bc * 2;                                     /* its first 8x8 unit */
    const svt_lf_mode_info b0 = P.mi[ub_r * P.mi_stride + ub_c];
    const bool big = b0.sb_type == 9 && !(b0.is_inter && P.mixed) && ub_r + 4 <= P.mi_rows && ub_c + 4 <= P.mi_cols; /* a 32x32 block coded in this pass */
    if (big && !(cr == br && cc == bc + 1)) continue;                            /* its top-right cell's ticket codes it and sets the four flags */
    const int ur0 = cr * 2, uc0 = cc * 2;
#ifdef INTRA_PROF
    unsigned long long pt1 = __builtin_amdgcn_s_memtime();
#endif
    if (lane == 0) {
        /* (r, c - 1) and (r - 1, c); for the 32x32 block at (br, bc): its lowest left neighbour (br + 1, bc - 1) and its rightmost above
           neighbour (br - 1, bc + 1) -- the flags of the cells before them on their row / column were waited for by THEIR owners */
        const int wl_r = big ? br + 1 : cr, wl_c = (big ? bc : cc) - 1;
        const int wa_r = (big ? br : cr) - 1, wa_c = big ? bc + 1 : cc;
        if (wl_c >= 0) while (__hip_atomic_load(&done[wl_r * c_cols + wl_c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) __builtin_amdgcn_s_sleep(2);
        if (wa_r >= 0) while (__hip_atomic_load(&done[wa_r * c_cols + wa_c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) __builtin_amdgcn_s_sleep(2);
    }
    __syncthreads();
#ifdef INTRA_FENCES
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
#ifdef INTRA_PROF
    unsigned long long pt2 = __builtin_amdgcn_s_memtime();
#endif
    /* the 8x8 units of the cell (of the 32x32 block: only its first unit starts a block) in z-order */
    const int nu = big ? 1 : 4, u_r = big ? ub_r : ur0, u_c = big ? ub_c : uc0;
    for (int z = 0; z < nu; z++) {
        const int r = (z >> 1) & 1, c = z & 1;
        const int ur = u_r + r, uc = u_c + c;
        if (ur >= P.mi_rows || uc >= P.mi_cols) continue;
        const svt_lf_mode_info b = P.mi[ur * P.mi_stride + uc];
        const int sub = b.sb_type == 0; /* an 8x8 unit of four 4x4 luma blocks + one 4x4 chroma block per plane */
        const int w8 = b.sb_type == 3 || sub ? 1 : b.sb_type == 6 ? 2 : b.sb_type == 9 ? 4 : 0;
        if (b.is_inter && P.mixed) continue;
        if (w8 == 0 || b.is_inter) { if (lane == 0) atomicOr(P.status, 1); continue; }
        if ((ur % w8) || (uc % w8)) continue;
        if (w8 == 4 && !big) { if (lane == 0) atomicOr(P.status, 1); continue; } /* a 32x32 block where none can start (or one that leaves the picture): malformed */
        const int mode = plane ? b.pad_[2] : b.pad_[1];
        if (ur + w8 > P.mi_rows || uc + w8 > P.mi_cols || b.tx_size != (sub ? 0 : w8 == 1 ? 1 : w8 == 2 ? 2 : 3) || (!sub && mode > 9) || b.pad_[2] > 9) {
            if (lane == 0) atomicOr(P.status, 1);
            continue;
        }
        const int sb = (ur >> 3) * P.sb_cols + (uc >> 3);
        const int x0 = plane ? uc * 4 : uc * 8, y0 = plane ? ur * 4 : ur * 8, nn = plane ? w8 * 4 : w8 * 8;
        int eob;
        if (sub && plane == 0) { /* blocks 0..3 in the reference's order, each with its own mode (nibbles of pad_[1], pad_[0]) */
            const int m4 = (int)b.pad_[1] | (int)b.pad_[0] << 8;
            eob = 0;
            for (int q4 = 0; q4 < 4; q4++) {
                const int mq = (m4 >> (4 * q4)) & 15;
                if (mq > 9) { if (lane == 0) atomicOr(P.status, 1); continue; }
                eob |= intra_block<4>(P, 0, x0 + 4 * (q4 & 1), y0 + 4 * (q4 >> 1), mq, sb, tile, edge, !(q4 & 1));
            }
        }
        else if (nn == 32) eob = intra_block<32>(P, plane, x0, y0, mode, sb, tile, edge);
        else if (nn == 16) eob = intra_block<16>(P, plane, x0, y0, mode, sb, tile, edge);
        else if (nn == 8) eob = intra_block<8>(P, plane, x0, y0, mode, sb, tile, edge);
        else eob = intra_block<4>(P, plane, x0, y0, mode, sb, tile, edge);
        if (eob && lane == 0) P.nz[ur * P.mi_stride + uc] = 1; /* the three planes of a block may all store the same 1 */
    }
#ifdef INTRA_PROF
    unsigned long long pt3 = __builtin_amdgcn_s_memtime();
#endif
    /* publish the cell (the four cells of a 32x32 block): its reconstruction reaches memory before the flag does */
#ifdef INTRA_FENCES
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
#else
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); /* every (write-through) store of this wave has completed */
    __syncthreads();
#endif
#ifdef INTRA_PROF
    unsigned long long pt4 = __builtin_amdgcn_s_memtime();
#endif
    if (lane == 0) {
        if (big) {
            _Pragma("unroll") for (int k = 0; k < 4; k++) {
                const int rr = br + (k >> 1), c2 = bc + (k & 1);
                if (rr < c_rows && c2 < c_cols) __hip_atomic_store(&done[rr * c_cols + c2], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else __hip_atomic_store(&done[cell], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#ifdef INTRA_PROF
    { const unsigned long long pt5 = __builtin_amdgcn_s_memtime(); pf[0] += pt1 - pt0; pf[1] += pt2 - pt1; pf[2] += pt3 - pt2; pf[3] += pt4 - pt3; pf[4] += pt5 - pt4; pf_n++; }
#endif
  }
}

/* stand-in decision for an intra picture (no claim of coding efficiency; the public API's callback replaces it): 16x16 blocks with
 * DC prediction, 8x8 where a 16x16 block would cross the picture edge */
__global__ __launch_bounds__(256) void svt_md_intra_default_kernel(svt_lf_mode_info *mi, int mi_stride, int mi_rows, int mi_cols, int filter_level) {
    const int u = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (u >= mi_rows * mi_cols) return;
    const int r = u / mi_cols, c = u % mi_cols;
    const int fit16 = (r & ~1) + 2 <= mi_rows && (c & ~1) + 2 <= mi_cols;
    svt_lf_mode_info m;
    m.sb_type = fit16 ? 6 : 3; m.tx_size = fit16 ? 2 : 1; m.skip = 0; m.is_inter = 0; m.filter_level = (uint8_t)filter_level;
    m.pad_[0] = m.pad_[1] = m.pad_[2] = 0;
    mi[r * mi_stride + c] = m;
}

} // namespace

/* internal launchers (declared in svt_ctx.h; the entry points are in encdec.hip) */
int32_t svt_intra_launch(svt_hip_ctx *ctx, const svt_encdec_picture *p, int32_t width, int32_t height, int32_t mi_stride, const svt_quant_tables *d_qtabs,
                         const int16_t *d_iscan, const uint32_t iscan_off[16], int32_t *d_sync, int32_t *d_status, int32_t mixed) {
    intra_pic_dev P;
    memset(&P, 0, sizeof P);
    P.src[0] = p->src.y; P.src[1] = p->src.u; P.src[2] = p->src.v;
    P.pred[0] = p->pred.y; P.pred[1] = p->pred.u; P.pred[2] = p->pred.v;
    P.rec[0] = p->recon.y; P.rec[1] = p->recon.u; P.rec[2] = p->recon.v;
    P.src_stride[0] = p->src.y_stride; P.src_stride[1] = p->src.uv_stride;
    P.pred_stride[0] = p->pred.y_stride; P.pred_stride[1] = p->pred.uv_stride;
    P.rec_stride[0] = p->recon.y_stride; P.rec_stride[1] = p->recon.uv_stride;
    P.mi = p->d_lf_mi; P.mi_stride = mi_stride; P.mi_rows = height >> 3; P.mi_cols = width >> 3;
    P.sb_cols = (width + 63) >> 6; P.sb_rows = (height + 63) >> 6; P.width = width; P.height = height;
    P.qtabs = d_qtabs; P.iscan = d_iscan;
    for (int i = 0; i < 16; i++) P.iscan_off[i] = iscan_off[i];
    P.qcoeff = p->d_qcoeff; P.dqcoeff = p->d_dqcoeff; P.eob_map = p->d_eob_map; P.nz = p->d_nz; P.sync = d_sync; P.status = d_status; P.mixed = mixed;
    const int n_cell = ((width + 15) >> 4) * ((height + 15) >> 4); /* 16x16 luma cells: the unit of the wavefront */
    HIP_TRY(hipMemsetAsync(d_sync, 0, (size_t)(2 + 3 * n_cell) * sizeof(int32_t), ctx->stream));
    /* (function-local statics with an initialiser: initialised once, thread-safely -- several contexts may launch from several threads) */
    static const int wg_per_cu = [] { const char *e = getenv("SVT_HIP_INTRA_WG_PER_CU"); return e && atoi(e) > 0 ? atoi(e) : 2; }(); /* two: a diagonal's tickets are then mostly held by workgroups already waiting at their flags (2160p, 16x16 DC: 4.77 -> 4.34 ms) */
    static const int wg_cap = [] { const char *e = getenv("SVT_HIP_INTRA_WGS"); return e && atoi(e) > 0 ? atoi(e) : 0; }(); /* deployment knob: fewer workgroups = a longer pass that leaves more of the device to what runs beside it (bench.py: 128) */
    int grid = ctx->cu_count * wg_per_cu;
    const int cap = ctx->intra_wgs > 0 ? ctx->intra_wgs : wg_cap; /* the context's setting (svt_hip_ctx_set_intra_workgroups) before the environment's */
    if (cap && grid > cap) grid = cap;
    if (grid > 3 * n_cell) grid = 3 * n_cell;
    hipLaunchKernelGGL(svt_intra_kernel, dim3(grid), dim3(64), 0, ctx->stream, P);
    HIP_TRY(hipGetLastError());
    return SVT_HIP_OK;
}

int32_t svt_md_intra_default_launch(svt_hip_ctx *ctx, svt_lf_mode_info *d_lf_mi, int32_t mi_stride, int32_t mi_rows, int32_t mi_cols, int32_t filter_level) {
    hipLaunchKernelGGL(svt_md_intra_default_kernel, dim3((mi_rows * mi_cols + 255) / 256), dim3(256), 0, ctx->stream, d_lf_mi, mi_stride, mi_rows, mi_cols, filter_level);
    HIP_TRY(hipGetLastError());
    return SVT_HIP_OK;
}
