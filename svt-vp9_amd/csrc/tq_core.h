/* tq_core.h -- the per-block body of the transform kernels: residual -> forward DCT / ADST -> quantiser (-> distortion, rate) ->
 * inverse transform -> reconstruction of ONE N x N block by N lanes of a wave, as perform_coding_loop does it
 * (Source/Lib/Codec/EbEncDecProcess.c:365-587; file comment of tq_kernel.hip).  Shared by the batch kernels (tq_kernel.hip: inter
 * pictures, prediction read from memory) and the intra kernel (intra_kernel.hip: prediction computed in registers from the
 * neighbours' reconstruction), so that there is one copy of every rounding rule.  Device code only; everything is internal linkage. */
#ifndef SVT_TQ_CORE_H
#define SVT_TQ_CORE_H
#include <hip/hip_runtime.h>
#include "svt_ctx.h"
#include "txfm1d.h"
#include "encdec_core.h"
#include "rate_core.h"

namespace {

/* Coefficient rate fused behind the quantiser (svt_hip_tq_rd_batch_device): what coeff_rate_estimate
 * (Codec/EbRateDistortionCost.c:55-172) returns for the block just quantised, from the coefficients while they are still
 * in registers / LDS -- perform_dist_rate_calc's (Codec/EbEncDecProcess.c:700-745) distortion + rate pair in one pass. */
struct tq_rate_args {
    const svt_rate_tables *T;   /* cost tables (device) */
    const int16_t         *scan; /* {scan, neighbors} tables, canonical layout [tx_size][tx_type] (device) */
    int32_t               *bits; /* out, per block */
};

/* Block lists built on the device (csrc/encdec.hip): the launch is sized for the list's capacity on the host, the actual offset and
 * number of blocks of this transform size are read from device memory: p[s] = first block of size s, p[4 + s] = how many. */
struct tq_dev_count {
    const int32_t *p; int s;
    /* pos != null: the list is position codes (encdec_core.h: svt_tq_pos), not descriptors -- a block's descriptor is rebuilt from its code,
     * this launch's transform size and the picture's geometry record at geom + picture * geom_stride (svt_tq_block_from_pos) */
    const uint32_t *pos; const uint8_t *geom; int geom_stride; const uint32_t *iscan_off; int sb_cols;
};

/* Workgroups are persistent and XCD-aware: workgroup w runs on XCD w & 7 (round-robin dispatch), and the groups of blocks it
 * walks are a contiguous eighth of the batch -- neighbouring blocks (which share 64-byte lines of the planes: a 4x4 block's
 * row is 4 bytes) are then fetched into ONE XCD's L2 instead of two. */
struct tq_walk { int per_xcd, first, step, base; };
__device__ __forceinline__ tq_walk tq_walk_of(int ngroups) {
    tq_walk w;
    w.per_xcd = (ngroups + 7) >> 3;
    w.base    = (int)(blockIdx.x & 7) * w.per_xcd;
    w.first   = (int)(blockIdx.x >> 3);
    w.step    = (int)(gridDim.x >> 3);
    return w;
}

/* all-reduce over the N (8, 16 or 32) lanes of a block with DPP instead of ds_bpermute shuffles (~90 cycles each and two address
 * instructions; the eob all-reduce sits on the critical path in front of the rate walk): after quad_perm [1,0,3,2] and [2,3,0,1]
 * the four lanes of a quad agree, row_half_mirror (i <-> 7 - i) then brings the other quad, row_mirror (i <-> 15 - i) the other
 * eight, and for 32 lanes one ds_swizzle (lane ^ 16) the other row. */
#define TQ_DPP(v, ctrl) __builtin_amdgcn_update_dpp(0, (int)(v), (ctrl), 0xf, 0xf, false)
template <int N> __device__ __forceinline__ int tq_lanes_max(int v) {
    int o;
    o = TQ_DPP(v, 0xB1); v = o > v ? o : v;
    o = TQ_DPP(v, 0x4E); v = o > v ? o : v;
    if constexpr (N >= 8) { o = TQ_DPP(v, 0x141); v = o > v ? o : v; }
    if constexpr (N >= 16) { o = TQ_DPP(v, 0x140); v = o > v ? o : v; }
    if constexpr (N >= 32) { o = __builtin_amdgcn_ds_swizzle(v, 0x401F); v = o > v ? o : v; }
    return v;
}
template <int N> __device__ __forceinline__ uint32_t tq_lanes_sum(uint32_t v) {
    v += (uint32_t)TQ_DPP(v, 0xB1);
    v += (uint32_t)TQ_DPP(v, 0x4E);
    if constexpr (N >= 8) v += (uint32_t)TQ_DPP(v, 0x141);
    if constexpr (N >= 16) v += (uint32_t)TQ_DPP(v, 0x140);
    if constexpr (N >= 32) v += (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x401F);
    return v;
}

/* threads per workgroup: 256 for every instance.  Round 2 sized the fused-rate instances of the big transforms (64 / 128
 * threads) so that their LDS stayed under what ONE motion-estimation workgroup releases; with the cost slices (13.8 KB) and scan
 * tables (6 KB) per workgroup that left 5 (32x32) / 10 (16x16) waves per CU -- the stage was latency-bound by its own occupancy.
 * In the dependency-true step the transform stage's time on the GPU adds to the step 1:1, so the tables are shared by four
 * waves instead: 12 / 16 waves per CU, the stage alone 1.57 -> 1.32 ms per mini-GOP, the step 3.27 -> 3.17 ms. */
template <int N, bool RATE> constexpr int tq_threads() { return 256; }

template <int N> struct txcfg;
template <> struct txcfg<4> { static constexpr int size = SVT_TX_4X4, shift = 4; };
template <> struct txcfg<8> { static constexpr int size = SVT_TX_8X8, shift = 5; };
template <> struct txcfg<16> { static constexpr int size = SVT_TX_16X16, shift = 6; };
template <> struct txcfg<32> { static constexpr int size = SVT_TX_32X32, shift = 6; };

/* forward 1-D for the hybrid (vp9_dct.c) path: DCT outputs are stored to int16 */
template <int N> __device__ __forceinline__ void fwd1d_hybrid(const int32_t *v, int32_t *o, bool adst) {
    if constexpr (N == 4) {
        if (adst) tx_fadst4(v, o);
        else { tx_fdct4(v, o); _Pragma("unroll") for (int i = 0; i < 4; i++) o[i] = (int16_t)o[i]; }
    } else if constexpr (N == 8) {
        if (adst) tx_adst8(v, o);
        else { tx_fdct8(v, o, 1); _Pragma("unroll") for (int i = 0; i < 8; i++) o[i] = (int16_t)o[i]; }
    } else {
        /* each branch fills its own array and the results meet in selects: with one shared output array the compiler
         * turns the two-way merge into an indexed private-memory (scratch) access */
        int32_t oa[16], od[16];
        if (adst) tx_adst16(v, oa);
        else { tx_fdct16(v, od); _Pragma("unroll") for (int i = 0; i < 16; i++) od[i] = (int16_t)od[i]; }
        _Pragma("unroll") for (int i = 0; i < 16; i++) o[i] = adst ? oa[i] : od[i];
    }
}
/* forward 1-D for the DCT_DCT (fwd_txfm.c) path */
template <int N> __device__ __forceinline__ void fwd1d_dct(const int32_t *v, int32_t *o) {
    if constexpr (N == 4) tx_fdct4(v, o);
    else if constexpr (N == 8) tx_fdct8(v, o, 0);
    else if constexpr (N == 16) tx_fdct16(v, o);
    else tx_fdct32(v, o);
}
template <int N> __device__ __forceinline__ void inv1d(const int32_t *v, int32_t *o, bool adst) {
    if constexpr (N == 4) { if (adst) tx_iadst4(v, o); else tx_idct4(v, o); }
    else if constexpr (N == 8) { if (adst) tx_adst8(v, o); else tx_idct8(v, o); }
    else if constexpr (N == 16) {
        int32_t oa[16], od[16]; /* see fwd1d_hybrid */
        if (adst) tx_adst16(v, oa); else tx_idct16(v, od);
        _Pragma("unroll") for (int i = 0; i < 16; i++) o[i] = adst ? oa[i] : od[i];
    }
    else tx_idct32(v, o);
}

/* one N-byte row as N/4 dwords: a single 8/16-byte access (two for N = 32) when the row is aligned to it, dwords otherwise */
template <int N> __device__ __forceinline__ void row_load(const uint8_t *p, bool vec, uint32_t *w) {
    if constexpr (N == 4) { w[0] = *(const uint32_t *)p; }
    else if constexpr (N == 8) {
        if (vec) { const uint2 t = *(const uint2 *)p; w[0] = t.x; w[1] = t.y; }
        else { w[0] = ((const uint32_t *)p)[0]; w[1] = ((const uint32_t *)p)[1]; }
    } else {
        if (vec) { _Pragma("unroll") for (int j = 0; j < N / 16; j++) { const uint4 t = ((const uint4 *)p)[j]; w[4 * j] = t.x; w[4 * j + 1] = t.y; w[4 * j + 2] = t.z; w[4 * j + 3] = t.w; } }
        else { _Pragma("unroll") for (int j = 0; j < N / 4; j++) w[j] = ((const uint32_t *)p)[j]; }
    }
}
template <int N> __device__ __forceinline__ void row_store(uint8_t *p, bool vec, const uint32_t *w) {
    if constexpr (N == 4) { *(uint32_t *)p = w[0]; }
    else if constexpr (N == 8) {
        if (vec) *(uint2 *)p = make_uint2(w[0], w[1]);
        else { ((uint32_t *)p)[0] = w[0]; ((uint32_t *)p)[1] = w[1]; }
    } else {
        if (vec) { _Pragma("unroll") for (int j = 0; j < N / 16; j++) ((uint4 *)p)[j] = make_uint4(w[4 * j], w[4 * j + 1], w[4 * j + 2], w[4 * j + 3]); }
        else { _Pragma("unroll") for (int j = 0; j < N / 4; j++) ((uint32_t *)p)[j] = w[j]; }
    }
}

/* Between the passes of a block its N lanes exchange rows and columns through the block's LDS tile.  N <= 32: the lanes of a block sit in
 * ONE wave, and the LDS executes the accesses of a wave in the order they were issued -- so what has to be kept is the order of the
 * instructions, not a rendezvous of the workgroup's four waves.  (Until round 5 these were __syncthreads(): the 32x32 instance -- 13 % of
 * the stage's vector instructions -- took 38 % of its time, its waves waiting for each other five times per block.) */
__device__ __forceinline__ void tq_block_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ int clamp16(int v) { return v < -32768 ? -32768 : v > 32767 ? 32767 : v; }
__device__ __forceinline__ uint8_t clip_add(int d, int t) { int v = d + t; return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }

/* One block.  k: its descriptor; active: false for the idle slots of a partly filled workgroup (they run along, store nothing);
 * i: the lane's column (first pass) / row (second pass); t: the block's N x (N + 1) dword LDS tile; srow / prow: row i of source and
 * prediction, packed; eob_slot / dist_slot / bits_slot: where the block's results go (dist_slot may be null; bits_slot, rate_T, s_tc,
 * s_scan only with RATE); recon_base + k.recon_off = the block's reconstruction; dqcoeff may be null (the dequantised coefficients then
 * go from the quantiser to the inverse transform in registers and nowhere else).  Every lane of a wave has to call it (the block's lanes
 * meet in tq_block_sync), with k.do_recon uniform over the wave.  Returns the block's eob (every lane of the block). */
/* WT: the reconstruction is stored write-through (agent-scope dword stores), for a caller whose neighbour blocks are read by other
 * workgroups of the SAME launch (intra_kernel.hip): they then see it without a release fence -- on this part an agent-scope release /
 * acquire pair writes back / invalidates the whole L2 of the XCD, for every kernel running beside the caller */
template <int N, bool RATE, bool DIST, bool WT = false>
__device__ __forceinline__ int tq_block_body(const svt_tq_block &k, const bool active, const int i, int32_t *t, const uint32_t (&srow)[N / 4],
                                              const uint32_t (&prow)[N / 4], const svt_quant_tables *__restrict__ qtabs, const int16_t *__restrict__ iscan_all,
                                              int16_t *__restrict__ qcoeff, int16_t *__restrict__ dqcoeff, uint16_t *eob_slot, uint64_t *dist_slot, int32_t *bits_slot,
                                              const svt_rate_tables *rate_T, const uint32_t *s_tc, const int16_t *s_scan, uint8_t *recon_base) {
    constexpr int LS = N + 1;            /* padded LDS row stride in dwords */
    constexpr int SCAN_T = 3 * N * N + 2;
    const int  tx_type = (N == 32) ? SVT_DCT_DCT : k.tx_type;
    const bool col_adst = tx_type == SVT_ADST_DCT || tx_type == SVT_ADST_ADST;
    const bool row_adst = tx_type == SVT_DCT_ADST || tx_type == SVT_ADST_ADST;

    int32_t  v[N], o[N];
    /* ---- residual: row i of source and prediction (packed dwords) -> the residual row goes through the LDS tile and comes back as
     * COLUMN i ---- */
    _Pragma("unroll") for (int cc = 0; cc < N; cc++)
        t[i * LS + cc] = (int16_t)((int)((srow[cc >> 2] >> (8 * (cc & 3))) & 0xff) - (int)((prow[cc >> 2] >> (8 * (cc & 3))) & 0xff));
    /* the N lanes of a block sit in one wave and LDS accesses of a wave are ordered: no barrier needed here */
    _Pragma("unroll") for (int r = 0; r < N; r++) v[r] = t[r * LS + i];
    if (tx_type == SVT_DCT_DCT) {
        _Pragma("unroll") for (int r = 0; r < N; r++) v[r] *= (N == 4 ? 16 : 4);
        if (N == 4 && i == 0 && v[0]) ++v[0];
        fwd1d_dct<N>(v, o);
        _Pragma("unroll") for (int kk = 0; kk < N; kk++) {
            int32_t m;
            if constexpr (N == 32) m = (o[kk] + 1 + (o[kk] > 0)) >> 2;
            else m = (int16_t)o[kk];
            if (N == 32 && k.partial32 && kk >= 16) m = 0;
            t[kk * LS + i] = m;
        }
    } else {
        _Pragma("unroll") for (int r = 0; r < N; r++) v[r] = (int16_t)(v[r] * (N == 4 ? 16 : 4));
        if (N == 4 && i == 0 && v[0]) v[0] = (int16_t)(v[0] + 1);
        if constexpr (N < 32) fwd1d_hybrid<N>(v, o, col_adst);
        _Pragma("unroll") for (int kk = 0; kk < N; kk++)
            t[kk * LS + i] = (N == 16) ? (int16_t)((o[kk] + 1 + (o[kk] < 0)) >> 2) : (int16_t)o[kk];
    }
    tq_block_sync();
    /* ---- row transform (row i = vertical frequency i) ---- */
    _Pragma("unroll") for (int kk = 0; kk < N; kk++) v[kk] = t[i * LS + kk];
    int32_t c[N]; /* coefficients of row i */
    if (tx_type == SVT_DCT_DCT) {
        if constexpr (N == 16) { _Pragma("unroll") for (int kk = 0; kk < N; kk++) v[kk] = (v[kk] + 1) >> 2; }
        fwd1d_dct<N>(v, o);
        _Pragma("unroll") for (int kk = 0; kk < N; kk++) {
            if constexpr (N == 4) c[kk] = (int16_t)(((int16_t)o[kk] + 1) >> 2);
            else if constexpr (N == 8) c[kk] = (int16_t)((int16_t)o[kk] / 2);
            else if constexpr (N == 16) c[kk] = (int16_t)o[kk];
            else c[kk] = (int16_t)((o[kk] + 1 + (o[kk] < 0)) >> 2);
        }
        if (N == 32 && k.partial32) {
            _Pragma("unroll") for (int kk = 0; kk < N; kk++) if (i >= 16 || kk >= 16) c[kk] = 0;
        }
    } else {
        if constexpr (N < 32) fwd1d_hybrid<N>(v, o, row_adst);
        _Pragma("unroll") for (int kk = 0; kk < N; kk++) {
            const int32_t tt = o[kk];
            c[kk] = (N == 4) ? (int16_t)((tt + 1) >> 2) : (N == 8) ? (int16_t)((tt + (tt < 0)) >> 1) : (int16_t)tt;
        }
    }
    /* ---- quantise row i; eob = 1 + max scan position of a non-zero level ---- */
    svt_quant_tables q;
    {   /* the 20-byte table as five dwords instead of ten halfword loads */
        const uint32_t *qp = (const uint32_t *)(qtabs + k.qtab);
        uint32_t        qw5[5];
        _Pragma("unroll") for (int j = 0; j < 5; j++) qw5[j] = qp[j];
        q.zbin[0] = (int16_t)qw5[0]; q.zbin[1] = (int16_t)(qw5[0] >> 16); q.round[0] = (int16_t)qw5[1]; q.round[1] = (int16_t)(qw5[1] >> 16);
        q.quant[0] = (int16_t)qw5[2]; q.quant[1] = (int16_t)(qw5[2] >> 16); q.quant_shift[0] = (int16_t)qw5[3]; q.quant_shift[1] = (int16_t)(qw5[3] >> 16);
        q.dequant[0] = (int16_t)qw5[4]; q.dequant[1] = (int16_t)(qw5[4] >> 16);
    }
    /* row i of the inverse scan, fetched as dwords up front (its latency hides behind the transforms) */
    uint32_t isw[N / 2];
    {
        const uint32_t *ip = (const uint32_t *)(iscan_all + k.iscan_off + i * N);
        _Pragma("unroll") for (int q = 0; q < N / 2; q++) isw[q] = ip[q];
    }
    int eob = 0;
    uint32_t rdist = 0, pdist = 0;
    int32_t dq[N];
    /* row i of qcoeff / dqcoeff leaves as packed dwords: 2N bytes per lane, consecutive lanes consecutive rows */
    constexpr int VC = N == 4 ? 4 : 8; /* coefficients per vector store */
    uint32_t qw[VC / 2], dqw[VC / 2];
    int16_t *qo = qcoeff + k.coeff_off + i * N, *dqo = dqcoeff + k.coeff_off + i * N;
    _Pragma("unroll") for (int kk = 0; kk < N; kk++) {
        const int ac = (i | kk) != 0, cv = c[kk], sign = cv >> 31;
        int       a = (cv ^ sign) - sign, level = 0, qv = 0, dv = 0;
        if constexpr (N < 32) {
            if (a >= q.zbin[ac]) {
                const int tmp = clamp16(a + q.round[ac]);
                level = ((((tmp * q.quant[ac]) >> 16) + tmp) * q.quant_shift[ac]) >> 16;
                qv = (int16_t)((level ^ sign) - sign);
                dv = (int16_t)(qv * q.dequant[ac]);
            }
        } else {
            const int zbin = (q.zbin[ac] + 1) >> 1;
            if (cv >= zbin || cv <= -zbin) {
                a = clamp16(a + ((q.round[ac] + 1) >> 1));
                level = ((((a * q.quant[ac]) >> 16) + a) * q.quant_shift[ac]) >> 15;
                qv = (int16_t)((level ^ sign) - sign);
                dv = (int16_t)(qv * q.dequant[ac] / 2);
            }
        }
        dq[kk] = dv;
        /* full_distortion_kernel32bit (C_DEFAULT/EbPictureOperators_C.c:288-311): the difference goes through an int16_t
           parameter, the sums wrap in uint32_t */
        if constexpr (DIST) {
            const int dd = (int16_t)(cv - dv);
            rdist += (uint32_t)(dd * dd);
            pdist += (uint32_t)(cv * cv);
        }
        {   /* the row leaves in vectors of VC coefficients as soon as they are complete (few live registers) */
            const int j = (kk % VC) >> 1;
            if (kk & 1) { qw[j] |= (uint32_t)(uint16_t)qv << 16; dqw[j] |= (uint32_t)(uint16_t)dv << 16; }
            else { qw[j] = (uint16_t)qv; dqw[j] = (uint16_t)dv; }
            if ((kk % VC) == VC - 1) {
                if constexpr (N == 4) {
                    if (active) { *(uint2 *)qo = make_uint2(qw[0], qw[1]); if (dqcoeff) *(uint2 *)dqo = make_uint2(dqw[0], dqw[1]); }
                } else {
                    if (active) {
                        ((uint4 *)qo)[kk / VC]  = make_uint4(qw[0], qw[1], qw[2], qw[3]);
                        if (dqcoeff) ((uint4 *)dqo)[kk / VC] = make_uint4(dqw[0], dqw[1], dqw[2], dqw[3]); /* (uniform: a kernel argument) */
                    }
                    /* RATE: the block's quantised coefficients also go to the (now dead) transpose tile, raster order, for
                     * the scan walk below; the tile of a block is private to its N lanes, which sit in one wave */
                    if constexpr (RATE) ((uint4 *)t)[(i * N + kk - (VC - 1)) / 8] = make_uint4(qw[0], qw[1], qw[2], qw[3]);
                }
            }
        }
        if (level && active) { const int pos = (int)((isw[kk >> 1] >> (16 * (kk & 1))) & 0xffff) + 1; eob = pos > eob ? pos : eob; }
    }
    eob = tq_lanes_max<N>(eob);
    if (active && i == 0) *eob_slot = (uint16_t)eob;
    if constexpr (DIST) if (dist_slot) { /* T3: coefficient-domain distortion of the block, summed over its N lanes */
        rdist = tq_lanes_sum<N>(rdist); pdist = tq_lanes_sum<N>(pdist);
        if (active && i == 0) { dist_slot[0] = rdist; dist_slot[1] = pdist; }
    }
    if constexpr (RATE) {
        /* coeff_rate_estimate of the block: lane i takes the scan positions i, i + N, .. <= eob (rate_core.h) */
        const int      rinfo = k.pad_[0], ptype = (rinfo >> 2) & 1, inter = (rinfo >> 3) & 1, ctx0 = rinfo & 3;
        const int16_t *sc = s_scan + tx_type * SCAN_T; /* tx_type is 0 for 32x32 */
        int bits = rate_positions<N>((const int16_t *)t, sc, sc + N * N, s_tc + (ptype * 2 + inter) * RATE_SLICE, rate_T, i, eob, N * N,
                                     txcfg<N>::size, ctx0);
        bits = (int)tq_lanes_sum<N>((uint32_t)bits);
        if (active && i == 0) *bits_slot = bits;
    }
    if (k.do_recon) { /* uniform within a size group (launcher contract): the barriers below stay workgroup-uniform */
    /* ---- reconstruction: recon = pred + inverse transform (rows first, then columns) ---- */
    int32_t res[N];
    _Pragma("unroll") for (int r = 0; r < N; r++) res[r] = 0;
    const bool dct_path = tx_type == SVT_DCT_DCT;
    bool dc_only = false;
    int  nrows = N;
    if (dct_path) {
        if (N == 4) dc_only = eob <= 1;
        else dc_only = eob == 1;
        if (N == 8) nrows = eob <= 12 ? 4 : 8;
        if (N == 16) nrows = eob <= 10 ? 4 : eob <= 38 ? 8 : 16;
        if (N == 32) nrows = eob <= 34 ? 8 : eob <= 135 ? 16 : 32;
    }
    tq_block_sync(); /* tile is reused */
    if (eob != 0 && !dc_only) {
        if (i < nrows) inv1d<N>(dq, o, row_adst);
        else { _Pragma("unroll") for (int kk = 0; kk < N; kk++) o[kk] = 0; }
        _Pragma("unroll") for (int kk = 0; kk < N; kk++) t[i * LS + kk] = (int16_t)o[kk];
    } else if (eob != 0 && i == 0) {
        t[0] = dq[0];
    }
    tq_block_sync();
    if (eob != 0 && !dc_only) {
        _Pragma("unroll") for (int r = 0; r < N; r++) v[r] = t[r * LS + i];
        inv1d<N>(v, o, col_adst);
        _Pragma("unroll") for (int r = 0; r < N; r++) res[r] = ((int16_t)o[r] + (1 << (txcfg<N>::shift - 1))) >> txcfg<N>::shift;
    } else if (eob != 0) {
        /* eb_vp9_idct*_1_add_c: inv_txfm.c:174, 368, 784, 1241 */
        int32_t d = tx_rsw((int16_t)t[0] * TX_C16);
        d = tx_rsw(d * TX_C16);
        const int32_t a1 = (d + (1 << (txcfg<N>::shift - 1))) >> txcfg<N>::shift;
        _Pragma("unroll") for (int r = 0; r < N; r++) res[r] = a1;
    }
    /* residual column i -> LDS -> row i; reconstruction = clip(pred row + residual row), stored as dwords */
    _Pragma("unroll") for (int r = 0; r < N; r++) t[r * LS + i] = res[r];
    if (active && recon_base) { /* (null: the block named a reconstruction set the caller did not pass) */
        /* a batch may reconstruct into several buffers (reference pictures of different mini-GOPs): pad_[0] bits 4-6 name the one */
        uint8_t *d = recon_base + k.recon_off + (size_t)i * k.recon_stride;
        uint32_t rw[N / 4];
        _Pragma("unroll") for (int q = 0; q < N / 4; q++) {
            uint32_t w = 0;
            _Pragma("unroll") for (int b = 0; b < 4; b++)
                w |= (uint32_t)clip_add((int)((prow[q] >> (8 * b)) & 0xff), t[i * LS + 4 * q + b]) << (8 * b);
            rw[q] = w;
        }
        if constexpr (WT) {
            _Pragma("unroll") for (int q = 0; q < N / 4; q++) __hip_atomic_store((uint32_t *)d + q, rw[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else row_store<N>(d, ((uintptr_t)d & (N >= 16 ? 15 : N - 1)) == 0, rw);
    }
    } /* do_recon */
    return eob;
}

} // namespace
#endif
