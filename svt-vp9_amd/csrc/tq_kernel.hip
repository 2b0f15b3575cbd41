/*
 * tq_kernel.hip -- residual -> forward DCT/ADST -> quantise/dequantise -> inverse transform + reconstruction
 * for batches of VP9 transform blocks (gfx950).
 *
 * Replaces the per-block body of perform_coding_loop (Source/Lib/Codec/EbEncDecProcess.c:365-587):
 *   eb_vp9_residual_kernel (C_DEFAULT/EbPictureOperators_C.c:204) -> eb_vp9_fdct32x32 / eb_vpx_partial_fdct32x32 /
 *   eb_vp9_fht16x16 / fht8x8 / fht4x4 / eb_vpx_fdct4x4 (VPX/fwd_txfm.c, VPX/vp9_dct.c) -> eb_vp9_quantize_b[_32x32]
 *   (VPX/quantize.c:112-254) -> pic_copy + eb_vp9_idct*_add / eb_vp9_iht*_add (VPX/vp9_idct.c:111-189, VPX/inv_txfm.c).
 *
 * Mapping: one N-point 1-D transform per lane, N lanes per NxN block, 64/N blocks per wave64 and 256/N per
 * workgroup; the column pass keeps a whole column in VGPRs, the transposition between the passes goes through a
 * padded (N+1 dwords per row) LDS tile so both passes are bank-conflict free; the eob is a max-reduction of
 * iscan positions over the block's N lanes (DPP/shuffle).  Integer butterflies on VALU -- no MFMA (these are
 * 32-bit integer rotations with data-dependent rounding/truncation, not dense contractions).
 * Source, prediction and reconstruction move as dword rows (lane i owns row i); the row <-> column changes go through the
 * same LDS tile; coefficient rows leave as 8/16-byte vectors (coeff_off and the two coefficient arrays must be 16-byte
 * aligned: every block starts on a multiple of 8 coefficients).  Global traffic per block: N*N source + N*N prediction bytes in, 2*N*N int16 (qcoeff, dqcoeff) + N*N recon out.
 */
#include <hip/hip_runtime.h>
#include "tq_core.h"
#include <stdlib.h>

namespace {

/* 32x32: 54 KB of LDS per 256-thread workgroup (eight blocks + the rate tables) allow three workgroups = 12 waves per CU */
template <int N, bool RATE, bool DIST>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(N == 32 ? (RATE ? 3 : 4) : 1))) void svt_tq_kernel(const uint8_t *__restrict__ src, const uint8_t *__restrict__ pred,
                                                     uint8_t *__restrict__ recon, const svt_tq_block *__restrict__ blocks,
                                                     int n_blocks, const svt_quant_tables *__restrict__ qtabs,
                                                     const int16_t *__restrict__ iscan_all, int16_t *__restrict__ qcoeff,
                                                     int16_t *__restrict__ dqcoeff, uint16_t *__restrict__ eob_out,
                                                     uint64_t *__restrict__ dist_out, tq_rate_args ra, const uint8_t *const *__restrict__ recon_set, tq_dev_count dc) {
    if (dc.p) { /* device-built list: where this size's blocks start and how many there are */
        const int o = dc.p[dc.s];
        n_blocks = dc.p[4 + dc.s];
        blocks += o; eob_out += o;
        if (dc.pos) dc.pos += o;
        if (dist_out) dist_out += 2 * o;
        if constexpr (RATE) ra.bits += o;
    }
    constexpr int NT  = tq_threads<N, RATE>();
    constexpr int BPW = NT / N;           /* blocks per workgroup */
    constexpr int LS  = N + 1;            /* padded LDS row stride in dwords */
    __shared__ int32_t tile[BPW][N * LS];
    /* RATE: the four token-cost slices [plane_type][is_inter] of this transform size, copied once per (persistent) workgroup */
    __shared__ uint32_t s_tc[RATE ? 4 * RATE_SLICE : 1];
    /* ... and the {scan, neighbours} tables of this size (all four transform types; one for 32x32): the walk's three table
     * reads per position then cost an LDS access instead of a dependent global round trip */
    /* (the two entries that close a table -- the neighbours of position n -- are never read: a walk ends at position eob <= n - 1 or,
     * for a full block, without an EOB token; leaving them out of the single 32x32 table keeps that instance at 53 760 bytes = 42 LDS
     * granules, three workgroups per CU instead of two) */
    constexpr int SCAN_T = 3 * N * N + 2, SCAN_N = N == 32 ? 3 * N * N : 4 * SCAN_T;
    __shared__ int16_t s_scan[RATE ? SCAN_N : 2];
    if constexpr (RATE) {
        const uint32_t *g = &ra.T->token_costs[txcfg<N>::size][0][0][0][0][0][0];
        for (int j = threadIdx.x; j < 4 * RATE_SLICE; j += NT) s_tc[j] = g[j];
        const uint32_t *gs = (const uint32_t *)(ra.scan + rate_scan_offset(txcfg<N>::size, 0)); /* 4-byte aligned, even count */
        for (int j = threadIdx.x; j < SCAN_N / 2; j += NT) ((uint32_t *)s_scan)[j] = gs[j];
        __syncthreads();
    }
    const int lb  = threadIdx.x / N;      /* block slot inside the workgroup */
    const int i   = threadIdx.x % N;      /* column (pass 1) / row (pass 2) owned by this lane */
    const tq_walk wk = tq_walk_of((n_blocks + BPW - 1) / BPW);
  for (int gj = wk.first; gj < wk.per_xcd; gj += wk.step) {
    const int grp = wk.base + gj;
    if (grp * BPW >= n_blocks) break;     /* uniform: depends on the workgroup only */
    const int blk = grp * BPW + lb;
    const bool active = blk < n_blocks;
    svt_tq_block k;
    if (dc.pos) { /* (uniform) the list holds position codes: the descriptor is rebuilt in registers */
        const uint32_t pc = dc.pos[active ? blk : 0];
        svt_tq_block_from_pos(pc, txcfg<N>::size, (const svt_tq_pic_geom *)(dc.geom + (size_t)svt_tq_pos_pic(pc) * dc.geom_stride), dc.iscan_off, dc.sb_cols, &k);
    } else if (active) k = blocks[blk];
    else { k = blocks[0]; }
    int32_t *t = tile[lb];
    /* lane i fetches ROW i of source and prediction as dwords (coalesced: the N lanes of a block read N consecutive rows of N bytes) */
    uint32_t srow[N / 4], prow[N / 4];
    {
        const uint8_t *sp = src + k.src_off + (size_t)i * k.src_stride;
        const uint8_t *pp = pred + k.pred_off + (size_t)i * k.pred_stride;
        constexpr uintptr_t AM = N >= 16 ? 15 : N - 1;
        _Pragma("unroll") for (int q = 0; q < N / 4; q++) { srow[q] = 0u; prow[q] = 0u; }
        if (active) { row_load<N>(sp, ((uintptr_t)sp & AM) == 0, srow); row_load<N>(pp, ((uintptr_t)pp & AM) == 0, prow); }
    }
    /* a batch may reconstruct into several buffers (reference pictures of different mini-GOPs): pad_[0] bits 4-6 name the one */
    int32_t *bits_slot = nullptr;
    if constexpr (RATE) bits_slot = ra.bits + blk;
    tq_block_body<N, RATE, DIST>(k, active, i, t, srow, prow, qtabs, iscan_all, qcoeff, dqcoeff, eob_out + blk, dist_out ? dist_out + 2 * blk : nullptr, bits_slot,
                                 ra.T, s_tc, s_scan, recon_set ? (uint8_t *)recon_set[(k.pad_[0] >> 4) & 7] : recon);
    tq_block_sync(); /* the block's tile is rewritten by its lanes' next block */
  }
}


/* ---- coefficient rate of a 4x4 block inside the lane that owns it ----
 * The three 4x4 scan orders of VP9 (VPX/vp9_scan.c: default_scan_4x4 for DCT_DCT and ADST_ADST, row_scan_4x4 for ADST_DCT,
 * col_scan_4x4 for DCT_ADST) with their neighbour pairs are compile-time tables here, so that the unrolled walk over the
 * scan positions addresses the lane's 16 token / energy registers with constant indices: no LDS staging of the coefficients,
 * no cross-lane step.  svt_hip_rate_scan4x4_table exposes them; tests/test_rate.py checks them against the reference's tables
 * (the scan array the caller passes must hold the same ones -- they are normative).  */
struct scan4_tab { uint8_t scan[16], nb[32]; };
constexpr scan4_tab SCAN4[3] = {
    {{0, 4, 1, 5, 8, 2, 12, 9, 3, 6, 13, 10, 7, 14, 11, 15},
     {0, 0, 0, 0, 0, 0, 1, 4, 4, 4, 1, 1, 8, 8, 5, 8, 2, 2, 2, 5, 9, 12, 6, 9, 3, 6, 10, 13, 7, 10, 11, 14}},
    {{0, 1, 4, 2, 5, 3, 6, 8, 9, 7, 12, 10, 13, 11, 14, 15},
     {0, 0, 0, 0, 0, 0, 1, 1, 4, 4, 2, 2, 5, 5, 4, 4, 8, 8, 6, 6, 8, 8, 9, 9, 12, 12, 10, 10, 13, 13, 14, 14}},
    {{0, 4, 8, 1, 12, 5, 9, 2, 13, 6, 10, 3, 7, 14, 11, 15},
     {0, 0, 0, 0, 4, 4, 0, 0, 8, 8, 1, 1, 5, 5, 1, 1, 9, 9, 2, 2, 6, 6, 2, 2, 3, 3, 10, 10, 7, 7, 11, 11}}};
constexpr uint8_t BAND4[16] = {0, 1, 1, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 5, 5, 5};
__host__ __device__ constexpr int scan4_kind(int tx_type) { return tx_type == SVT_ADST_DCT ? 1 : tx_type == SVT_DCT_ADST ? 2 : 0; }

/* tok[r] / en[r]: token and energy class of raster coefficient r; tc: the block's cost slice in LDS.  Position c contributes
 * tc[band][previous token == ZERO][ctx][token] while c < eob and the EOB token at c == eob (< 16); the position-independent
 * value costs are added by the caller.  The loop is unrolled; it leaves as soon as no lane of the wave has c <= eob. */
template <int K>
__device__ __forceinline__ int rate_walk4(const int (&tok)[16], const int (&en)[16], const uint32_t *tc, int eob, int ctx0) {
    int sum = (int)tc[ctx0 * 12 + (eob == 0 ? 11 : tok[0])];
    _Pragma("unroll") for (int c = 1; c < 16; c++) {
        if (!__any(c <= eob)) break;
        const int  pt = (1 + en[SCAN4[K].nb[2 * c]] + en[SCAN4[K].nb[2 * c + 1]]) >> 1;
        const bool in = c < eob;
        const int  pz = in && tok[SCAN4[K].scan[c - 1]] == 0, tk = in ? tok[SCAN4[K].scan[c]] : 11;
        const int  v  = (int)tc[((BAND4[c] * 2 + pz) * 6 + pt) * 12 + tk];
        sum += c <= eob ? v : 0;
    }
    return sum;
}

/* 4x4 (and, unused, 8x8) blocks: one block per lane.  With N lanes per block the small transforms are a sliver of the instruction
 * stream (descriptor decode and table loads repeated in every lane, LDS transposes, shuffles for eob and distortion: ~125
 * VALU lane-operations per sample, as many as a 32x32 block); a lane that owns the whole block keeps its samples in
 * registers from the residual to the reconstruction, needs no LDS, no barrier and no cross-lane step.  Same arithmetic as
 * svt_tq_kernel<N>.  4x4: the fwd_txfm.c and vp9_dct.c forms of the DCT coincide (the int16 casts are no-ops at this size);
 * 8x8: they differ in the cast inside tx_fdct8 (vp9_dct.c:67-68), selected by the block's transform type.  The inverse
 * always runs all rows (the reference's reduced variants are shortcuts with identical results).  Memory instructions per
 * block are unchanged (a lane issues the N row loads its N lanes issued). */
/* one 4x4 (8x8) block, whole, in the calling lane: the body of svt_tq_lane_kernel and of the 4x4 part of svt_tq_sb_kernel */
template <int N, bool RATE, bool DIST>
__device__ __forceinline__ void tq_lane_block(const svt_tq_block &k, const int blk, const uint8_t *__restrict__ src, const uint8_t *__restrict__ pred, uint8_t *__restrict__ recon,
                                              const svt_quant_tables *__restrict__ qtabs, const int16_t *__restrict__ iscan_all, int16_t *__restrict__ qcoeff,
                                              int16_t *__restrict__ dqcoeff, uint16_t *__restrict__ eob_out, uint64_t *__restrict__ dist_out, const tq_rate_args &ra,
                                              const uint32_t *s_tc, const int32_t *s_vc, const uint8_t *const *__restrict__ recon_set) {
    constexpr int ND = N / 4; /* dwords per row of samples */
    const bool col_adst = k.tx_type == SVT_ADST_DCT || k.tx_type == SVT_ADST_ADST;
    const bool row_adst = k.tx_type == SVT_DCT_ADST || k.tx_type == SVT_ADST_ADST;
    const bool dct_dct  = k.tx_type == SVT_DCT_DCT;
    uint32_t   prow[N][ND];
    int32_t    m[N][N]; /* after the column pass: m[kk][cc] = vertical frequency kk of column cc */
    {
        uint32_t srow[N][ND];
        _Pragma("unroll") for (int r = 0; r < N; r++) {
            const uint8_t *sp = src + k.src_off + (size_t)r * k.src_stride, *pp = pred + k.pred_off + (size_t)r * k.pred_stride;
            if constexpr (N == 4) { srow[r][0] = *(const uint32_t *)sp; prow[r][0] = *(const uint32_t *)pp; }
            else {
                row_load<N>(sp, ((uintptr_t)sp & 7) == 0, srow[r]);
                row_load<N>(pp, ((uintptr_t)pp & 7) == 0, prow[r]);
            }
        }
        _Pragma("unroll") for (int cc = 0; cc < N; cc++) {
            int32_t v[N], o[N];
            _Pragma("unroll") for (int r = 0; r < N; r++)
                v[r] = ((int)((srow[r][cc >> 2] >> (8 * (cc & 3))) & 0xff) - (int)((prow[r][cc >> 2] >> (8 * (cc & 3))) & 0xff)) * (N == 4 ? 16 : 4);
            if constexpr (N == 4) {
                if (cc == 0 && v[0]) ++v[0];
                if (col_adst) tx_fadst4(v, o); else tx_fdct4(v, o);
            } else {
                if (col_adst) tx_adst8(v, o);
                else if (dct_dct) tx_fdct8(v, o, 0);
                else tx_fdct8(v, o, 1);
            }
            _Pragma("unroll") for (int kk = 0; kk < N; kk++) m[kk][cc] = (int16_t)o[kk];
        }
    }
    /* quantiser tables of the block (lanes of one table read the same addresses) */
    svt_quant_tables q;
    {
        const uint32_t *qp = (const uint32_t *)(qtabs + k.qtab);
        uint32_t        qw5[5];
        _Pragma("unroll") for (int j = 0; j < 5; j++) qw5[j] = qp[j];
        q.zbin[0] = (int16_t)qw5[0]; q.zbin[1] = (int16_t)(qw5[0] >> 16); q.round[0] = (int16_t)qw5[1]; q.round[1] = (int16_t)(qw5[1] >> 16);
        q.quant[0] = (int16_t)qw5[2]; q.quant[1] = (int16_t)(qw5[2] >> 16); q.quant_shift[0] = (int16_t)qw5[3]; q.quant_shift[1] = (int16_t)(qw5[3] >> 16);
        q.dequant[0] = (int16_t)qw5[4]; q.dequant[1] = (int16_t)(qw5[4] >> 16);
    }
    /* ---- row pass, quantisation, distortion, eob; each row of coefficients leaves as one 8/16-byte vector ---- */
    int32_t  dq[N][N];
    uint32_t rdist = 0, pdist = 0;
    int      eob = 0;
    int      tok[RATE ? 16 : 1], en[RATE ? 16 : 1], vsum = 0, nnz = 0; /* RATE: per raster coefficient */
    const uint32_t *ip = (const uint32_t *)(iscan_all + k.iscan_off);
    _Pragma("unroll") for (int i = 0; i < N; i++) {
        int32_t  o[N];
        uint32_t isw[N / 2], qw[N / 2], dqw[N / 2];
        _Pragma("unroll") for (int j = 0; j < N / 2; j++) isw[j] = ip[i * (N / 2) + j];
        if constexpr (N == 4) {
            if (row_adst) tx_fadst4(m[i], o);
            else { tx_fdct4(m[i], o); _Pragma("unroll") for (int kk = 0; kk < N; kk++) o[kk] = (int16_t)o[kk]; }
        } else {
            if (row_adst) tx_adst8(m[i], o);
            else {
                if (dct_dct) tx_fdct8(m[i], o, 0); else tx_fdct8(m[i], o, 1);
                _Pragma("unroll") for (int kk = 0; kk < N; kk++) o[kk] = (int16_t)o[kk];
            }
        }
        _Pragma("unroll") for (int kk = 0; kk < N; kk++) {
            const int cv = N == 4 ? (int16_t)((o[kk] + 1) >> 2) : (int16_t)((o[kk] + (o[kk] < 0)) >> 1);
            const int ac = (i | kk) != 0, sign = cv >> 31, a = (cv ^ sign) - sign;
            int       level = 0, qv = 0, dv = 0;
            if (a >= q.zbin[ac]) {
                const int tmp = clamp16(a + q.round[ac]);
                level = ((((tmp * q.quant[ac]) >> 16) + tmp) * q.quant_shift[ac]) >> 16;
                qv = (int16_t)((level ^ sign) - sign);
                dv = (int16_t)(qv * q.dequant[ac]);
            }
            dq[i][kk] = dv;
            if constexpr (DIST) {
                const int dd = (int16_t)(cv - dv);
                rdist += (uint32_t)(dd * dd);
                pdist += (uint32_t)(cv * cv);
            }
            if (kk & 1) { qw[kk >> 1] |= (uint32_t)(uint16_t)qv << 16; dqw[kk >> 1] |= (uint32_t)(uint16_t)dv << 16; }
            else { qw[kk >> 1] = (uint16_t)qv; dqw[kk >> 1] = (uint16_t)dv; }
            if (level) { const int pos = (int)((isw[kk >> 1] >> (16 * (kk & 1))) & 0xffff) + 1; eob = pos > eob ? pos : eob; }
            if constexpr (RATE) {
                /* token, energy class and value cost of this coefficient; skipped when it is zero in every lane of the wave */
                tok[i * N + kk] = 0; en[i * N + kk] = 0;
                if (__any(level != 0)) {
                    const int al = qv < 0 ? -qv : qv; /* |level| as the reference sees it (the int16 qcoeff) */
                    tok[i * N + kk] = token_of(qv);
                    en[i * N + kk]  = energy_of_value(qv);
                    nnz += qv != 0;
                    int vc = s_vc[66 + (al < 67 ? qv : 0)];
                    if (__any(al >= 67)) { /* CAT6: vp9_get_token_cost, VPX/vp9_tokenize.h:118-127 */
                        const int extra = al >= 67 ? al - 67 : 0;
                        const int c6 = ra.T->cat6_low_cost[extra & 0xff] + ra.T->cat6_high_cost[extra >> 8];
                        vc = al >= 67 ? c6 : vc;
                    }
                    vsum += qv != 0 ? vc : 0;
                }
            }
        }
        int16_t *qo = qcoeff + k.coeff_off + i * N, *dqo = dqcoeff + k.coeff_off + i * N;
        if constexpr (N == 4) { *(uint2 *)qo = make_uint2(qw[0], qw[1]); if (dqcoeff) *(uint2 *)dqo = make_uint2(dqw[0], dqw[1]); }
        else { *(uint4 *)qo = make_uint4(qw[0], qw[1], qw[2], qw[3]); if (dqcoeff) *(uint4 *)dqo = make_uint4(dqw[0], dqw[1], dqw[2], dqw[3]); }
    }
    eob_out[blk] = (uint16_t)eob;
    if constexpr (DIST) if (dist_out) { dist_out[2 * blk] = rdist; dist_out[2 * blk + 1] = pdist; }
    if constexpr (RATE) {
        const int       rinfo = k.pad_[0], ptype = (rinfo >> 2) & 1, inter = (rinfo >> 3) & 1, ctx0 = rinfo & 3;
        const uint32_t *tc = s_tc + (ptype * 2 + inter) * RATE_SLICE;
        const int       kind = scan4_kind(k.tx_type);
        int             bits = 0;
        if (kind == 0) bits = rate_walk4<0>(tok, en, tc, eob, ctx0);
        if (kind == 1) bits = rate_walk4<1>(tok, en, tc, eob, ctx0);
        if (kind == 2) bits = rate_walk4<2>(tok, en, tc, eob, ctx0);
        /* value costs: every position before eob pays the cost of its value; the zeros among them pay value_cost[0 + 66] */
        ra.bits[blk] = bits + vsum + (eob - nnz) * s_vc[66];
    }
    if (!k.do_recon) return;
    /* ---- reconstruction: rows first, then columns (vp9_idct.c:111-189, inv_txfm.c); a column's eight results go straight
     * into the packed output rows ---- */
    constexpr int SH = txcfg<N>::shift;
    uint32_t      rw[N][ND];
    _Pragma("unroll") for (int r = 0; r < N; r++) { _Pragma("unroll") for (int j = 0; j < ND; j++) rw[r][j] = 0; }
    const bool dc_only = dct_dct && (N == 4 ? eob <= 1 : eob == 1);
    if (eob != 0 && !dc_only) {
        int32_t t[N][N];
        _Pragma("unroll") for (int i = 0; i < N; i++) {
            int32_t o[N];
            if constexpr (N == 4) { if (row_adst) tx_iadst4(dq[i], o); else tx_idct4(dq[i], o); }
            else { if (row_adst) tx_adst8(dq[i], o); else tx_idct8(dq[i], o); }
            _Pragma("unroll") for (int kk = 0; kk < N; kk++) t[i][kk] = (int16_t)o[kk];
        }
        _Pragma("unroll") for (int cc = 0; cc < N; cc++) {
            int32_t v[N], o[N];
            _Pragma("unroll") for (int r = 0; r < N; r++) v[r] = t[r][cc];
            if constexpr (N == 4) { if (col_adst) tx_iadst4(v, o); else tx_idct4(v, o); }
            else { if (col_adst) tx_adst8(v, o); else tx_idct8(v, o); }
            _Pragma("unroll") for (int r = 0; r < N; r++) {
                const int res = ((int16_t)o[r] + (1 << (SH - 1))) >> SH;
                rw[r][cc >> 2] |= (uint32_t)clip_add((int)((prow[r][cc >> 2] >> (8 * (cc & 3))) & 0xff), res) << (8 * (cc & 3));
            }
        }
    } else {
        int32_t a1 = 0;
        if (eob != 0) { /* eb_vp9_idct4x4_1_add_c / idct8x8_1_add_c, inv_txfm.c:174, 368 */
            int32_t d = tx_rsw((int16_t)dq[0][0] * TX_C16);
            d = tx_rsw(d * TX_C16);
            a1 = (d + (1 << (SH - 1))) >> SH;
        }
        _Pragma("unroll") for (int r = 0; r < N; r++) {
            _Pragma("unroll") for (int cc = 0; cc < N; cc++)
                rw[r][cc >> 2] |= (uint32_t)clip_add((int)((prow[r][cc >> 2] >> (8 * (cc & 3))) & 0xff), a1) << (8 * (cc & 3));
        }
    }
    uint8_t *const rbase = recon_set ? (uint8_t *)recon_set[(k.pad_[0] >> 4) & 7] : recon;
    if (!rbase) return; /* a set index beyond the caller's n_set: nowhere to reconstruct to (never buffer 0 by default) */
    _Pragma("unroll") for (int r = 0; r < N; r++) {
        uint8_t *d = rbase + k.recon_off + (size_t)r * k.recon_stride;
        if constexpr (N == 4) *(uint32_t *)d = rw[r][0];
        else row_store<N>(d, ((uintptr_t)d & 7) == 0, rw[r]);
    }
}

template <int N, bool RATE, bool DIST>
__global__ __launch_bounds__(256) void svt_tq_lane_kernel(const uint8_t *__restrict__ src, const uint8_t *__restrict__ pred,
                                                          uint8_t *__restrict__ recon, const svt_tq_block *__restrict__ blocks,
                                                          int n_blocks, const svt_quant_tables *__restrict__ qtabs,
                                                          const int16_t *__restrict__ iscan_all, int16_t *__restrict__ qcoeff,
                                                          int16_t *__restrict__ dqcoeff, uint16_t *__restrict__ eob_out,
                                                          uint64_t *__restrict__ dist_out, tq_rate_args ra, const uint8_t *const *__restrict__ recon_set, tq_dev_count dc) {
    if (dc.p) {
        const int o = dc.p[dc.s];
        n_blocks = dc.p[4 + dc.s];
        blocks += o; eob_out += o;
        if (dc.pos) dc.pos += o;
        if (dist_out) dist_out += 2 * o;
        if constexpr (RATE) ra.bits += o;
    }
    static_assert(N == 4 || N == 8, "block-per-lane form: 4x4 and 8x8 only");
    static_assert(!RATE || N == 4, "in-lane rate: 4x4 only");
    /* RATE: the four 4x4 token-cost slices [plane_type][is_inter] and the value-cost table, copied once per workgroup */
    __shared__ uint32_t s_tc[RATE ? 4 * RATE_SLICE : 1];
    __shared__ int32_t  s_vc[RATE ? 136 : 1];
    if constexpr (RATE) {
        const uint32_t *g = &ra.T->token_costs[0][0][0][0][0][0][0];
        for (int j = threadIdx.x; j < 4 * RATE_SLICE; j += 256) s_tc[j] = g[j];
        if (threadIdx.x < 133) s_vc[threadIdx.x] = ra.T->value_cost[threadIdx.x];
        __syncthreads();
    }
    const tq_walk wk = tq_walk_of((n_blocks + 255) / 256);
  for (int gj = wk.first; gj < wk.per_xcd; gj += wk.step) {
    const int blk = (wk.base + gj) * 256 + (int)threadIdx.x;
    if (blk >= n_blocks) continue; /* no barrier inside the loop */
    svt_tq_block k;
    if (dc.pos) {
        const uint32_t pc = dc.pos[blk];
        svt_tq_block_from_pos(pc, txcfg<N>::size, (const svt_tq_pic_geom *)(dc.geom + (size_t)svt_tq_pos_pic(pc) * dc.geom_stride), dc.iscan_off, dc.sb_cols, &k);
    } else k = blocks[blk];
    tq_lane_block<N, RATE, DIST>(k, blk, src, pred, recon, qtabs, iscan_all, qcoeff, dqcoeff, eob_out, dist_out, ra, s_tc, s_vc, recon_set);
  }
}

/* ---- the encode pass's transform stage as ONE launch over SB-ordered lists (round 6) ----
 * Four launches, one per transform size, each fetch the 128-byte lines of source and prediction they share with the others: an SB whose
 * 32x32 areas carry different transform sizes was read up to four times (2.08 GB per 16 pictures at 2160p against 1.0 GB of samples and
 * coefficients, profiles/r05_pmc_traffic.md) -- and at 0.54 ms for the four launches that is 3.8 TB/s: the stage was as close to the HBM
 * roof as to the issue roof.  Here the list is ordered [picture][chunk of SVT_TQ_CHUNK_SBS SBs][size][SB][unit][plane] (csrc/encdec.hip) and a
 * workgroup is (chunk, size, part): the SVT_TQ_SLOTS workgroups of a chunk are neighbours in dispatch order ON ONE XCD (workgroup w runs
 * on XCD w & 7), so a line one of them has fetched is in that XCD's L2 when the next one asks for it.  seg[] = the exclusive prefix of the
 * per-(picture, chunk, size, SB) counts inside a picture, bases[picture] (+ one closing entry) the pictures' first blocks.
 * 128 threads: four 32x32 blocks keep the transpose tiles at 16.9 KB (eight workgroups per CU at the 128 registers the 32x32 body needs). */
constexpr int TQ_SB_NT = 128;
__host__ __device__ constexpr int tq_sb_slots(int s) { return s == 0 ? 4 : s == 1 ? 4 : 2; } /* workgroups per (chunk, size) */
__host__ __device__ constexpr int tq_sb_slots_range(int lo, int hi) { int n = 0; for (int s = lo; s <= hi; s++) n += tq_sb_slots(s); return n; }

template <int N>
__device__ __forceinline__ void tq_sb_groups(int32_t *tile_mem, const int first, const int count, const int part, const int nparts, const uint8_t *__restrict__ src,
                                             const uint8_t *__restrict__ pred, const svt_quant_tables *__restrict__ qtabs, const int16_t *__restrict__ iscan_all,
                                             int16_t *__restrict__ qcoeff, int16_t *__restrict__ dqcoeff, uint16_t *__restrict__ eob_out,
                                             const uint8_t *const *__restrict__ recon_set, const tq_dev_count &dc) {
    constexpr int BPW = TQ_SB_NT / N, LS = N + 1;
    const int     lb = (int)threadIdx.x / N, i = (int)threadIdx.x % N;
    int32_t      *t = tile_mem + lb * (N * LS);
    for (int g = part; g * BPW < count; g += nparts) { /* (uniform) */
        const int  b = g * BPW + lb;
        const bool active = b < count;
        const int  blk = first + (active ? b : 0);
        svt_tq_block k;
        const uint32_t pc = dc.pos[blk];
        svt_tq_block_from_pos(pc, txcfg<N>::size, (const svt_tq_pic_geom *)(dc.geom + (size_t)svt_tq_pos_pic(pc) * dc.geom_stride), dc.iscan_off, dc.sb_cols, &k);
        uint32_t srow[N / 4], prow[N / 4];
        {
            const uint8_t *sp = src + k.src_off + (size_t)i * k.src_stride;
            const uint8_t *pp = pred + k.pred_off + (size_t)i * k.pred_stride;
            constexpr uintptr_t AM = N >= 16 ? 15 : N - 1;
            _Pragma("unroll") for (int q = 0; q < N / 4; q++) { srow[q] = 0u; prow[q] = 0u; }
            if (active) { row_load<N>(sp, ((uintptr_t)sp & AM) == 0, srow); row_load<N>(pp, ((uintptr_t)pp & AM) == 0, prow); }
        }
        tq_block_body<N, false, false>(k, active, i, t, srow, prow, qtabs, iscan_all, qcoeff, dqcoeff, eob_out + blk, nullptr, nullptr, nullptr, nullptr, nullptr,
                                       (uint8_t *)recon_set[(k.pad_[0] >> 4) & 7]);
        tq_block_sync();
    }
}

/* sizes S_LO .. S_HI of every chunk.  <0, 3>: everything in one launch (128 registers, 16.9 KB of LDS in EVERY workgroup, the 4x4 ones too);
 * <0, 2> + <3, 3>: the small sizes in a launch that needs 96 registers and 8.7 KB, the 32x32 blocks in one of their own -- see the launcher */
template <int S_LO, int S_HI>
__global__ __launch_bounds__(TQ_SB_NT) __attribute__((amdgpu_waves_per_eu(S_HI == 3 ? 4 : 5))) void svt_tq_sb_kernel(
    const uint8_t *__restrict__ src, const uint8_t *__restrict__ pred, const svt_quant_tables *__restrict__ qtabs, const int16_t *__restrict__ iscan_all,
    int16_t *__restrict__ qcoeff, int16_t *__restrict__ dqcoeff, uint16_t *__restrict__ eob_out, const uint8_t *const *__restrict__ recon_set, tq_dev_count dc,
    const int32_t *__restrict__ seg, const int32_t *__restrict__ bases, int n_chunks, int seg_per_chunk /* 4 * chunk SBs */, int n_items /* pictures x chunks */) {
    constexpr int NMAX = 4 << S_HI, SLOTS = tq_sb_slots_range(S_LO, S_HI);
    __shared__ int32_t tile_mem[S_HI == 0 ? 1 : (TQ_SB_NT / NMAX) * NMAX * (NMAX + 1)];
    /* workgroup -> (item, slot): the slots of an item are consecutive on one XCD */
    const int xcd = (int)(blockIdx.x & 7), local = (int)(blockIdx.x >> 3);
    const int item = (local / SLOTS) * 8 + xcd, slot = local % SLOTS;
    if (item >= n_items) return;
    int s = S_LO, part = slot;
    _Pragma("unroll") for (int q = S_LO; q < S_HI; q++) if (s == q && part >= tq_sb_slots(q)) { part -= tq_sb_slots(q); s = q + 1; }
    const int nparts = tq_sb_slots(s);
    const int pic = item / n_chunks, chunk = item - pic * n_chunks;
    const int per_pic = n_chunks * seg_per_chunk, at = chunk * seg_per_chunk + s * (seg_per_chunk >> 2), nxt = at + (seg_per_chunk >> 2);
    const int base = bases[pic];
    const int first = base + seg[pic * per_pic + at];
    const int last = nxt < per_pic ? base + seg[pic * per_pic + nxt] : bases[pic + 1];
    const int count = last - first;
    if (count <= 0) return;
    if (S_LO == 0 && s == 0) { /* 4x4: a block per lane, no LDS, no barrier */
        const tq_rate_args none = {nullptr, nullptr, nullptr};
        for (int b = part * TQ_SB_NT + (int)threadIdx.x; b < count; b += nparts * TQ_SB_NT) {
            const int      blk = first + b;
            const uint32_t pc = dc.pos[blk];
            svt_tq_block   k;
            svt_tq_block_from_pos(pc, SVT_TX_4X4, (const svt_tq_pic_geom *)(dc.geom + (size_t)svt_tq_pos_pic(pc) * dc.geom_stride), dc.iscan_off, dc.sb_cols, &k);
            tq_lane_block<4, false, false>(k, blk, src, pred, nullptr, qtabs, iscan_all, qcoeff, dqcoeff, eob_out, nullptr, none, nullptr, nullptr, recon_set);
        }
    }
    if (S_LO <= 1 && S_HI >= 1 && s == 1) tq_sb_groups<8>(tile_mem, first, count, part, nparts, src, pred, qtabs, iscan_all, qcoeff, dqcoeff, eob_out, recon_set, dc);
    if (S_LO <= 2 && S_HI >= 2 && s == 2) tq_sb_groups<16>(tile_mem, first, count, part, nparts, src, pred, qtabs, iscan_all, qcoeff, dqcoeff, eob_out, recon_set, dc);
    if (S_HI == 3 && s == 3) tq_sb_groups<32>(tile_mem, first, count, part, nparts, src, pred, qtabs, iscan_all, qcoeff, dqcoeff, eob_out, recon_set, dc);
}

/* persistent grid: a multiple of 8 workgroups (one walk per XCD, tq_walk_of), at most `per_cu` per compute unit */
int tq_grid(svt_hip_ctx *ctx, int ngroups, int per_cu) {
    const int cus = ctx->cu_count;
    static const int per_cu_env = getenv("SVT_HIP_TQ_PER_CU") ? atoi(getenv("SVT_HIP_TQ_PER_CU")) : 0;
    if (per_cu_env > 0) per_cu = per_cu_env;
    const int per_xcd = (ngroups + 7) / 8, cap = (cus * per_cu + 7) / 8;
    return 8 * (per_xcd < cap ? per_xcd : cap);
}

template <int N, bool RATE, bool DIST = true>
hipError_t launch_tq(svt_hip_ctx *ctx, hipStream_t st, const uint8_t *src, const uint8_t *pred, uint8_t *recon, const svt_tq_block *blocks, int n,
                     const svt_quant_tables *q, const int16_t *iscan, int16_t *qc, int16_t *dqc, uint16_t *eob, uint64_t *dist, tq_rate_args ra,
                     const uint8_t *const *recon_set, tq_dev_count dc = {nullptr, 0, nullptr, nullptr, 0, nullptr, 0}) {
    if (n <= 0) return hipSuccess;
    /* block per lane for 4x4 only: the 8x8 instance is bit-exact too but needs 201 VGPRs (64 samples + the transposed
     * intermediate live in one lane; 87 spills when held to 128) -- two waves per SIMD, and each displaces two ME waves:
     * the overlapped step went from 3.27 to 3.77 ms with it, so 8x8 stays on the N-lanes-per-block kernel */
    if constexpr (N == 4) {
        hipLaunchKernelGGL((svt_tq_lane_kernel<N, RATE, DIST>), dim3(tq_grid(ctx, (n + 255) / 256, 6)), dim3(256), 0, st, src, pred, recon, blocks, n, q,
                           iscan, qc, dqc, eob, dist, ra, recon_set, dc);
        return hipGetLastError();
    } else {
        constexpr int NT = tq_threads<N, RATE>(), BPW = NT / N;
        hipLaunchKernelGGL((svt_tq_kernel<N, RATE, DIST>), dim3(tq_grid(ctx, (n + BPW - 1) / BPW, 6)), dim3(NT), 0, st, src, pred, recon, blocks, n, q,
                           iscan, qc, dqc, eob, dist, ra, recon_set, dc);
        return hipGetLastError();
    }
}

template <bool RATE>
int32_t tq_launch_all(svt_hip_ctx *ctx, const uint8_t *d_src, const uint8_t *d_pred, uint8_t *d_recon, const svt_tq_block *d_blocks,
                      const int32_t size_count[4], const svt_quant_tables *d_qtabs, const int16_t *d_iscan, int16_t *d_qcoeff,
                      int16_t *d_dqcoeff, uint16_t *d_eob, uint64_t *d_dist, tq_rate_args ra, uint8_t *const *recon_set = nullptr, int n_set = 0) {
    HIP_TRY(hipSetDevice(ctx->device));
    /* several reconstruction buffers: their base pointers go to the device through the descriptor ring */
    const uint8_t *const *d_set = nullptr;
    if (recon_set) {
        void *h = nullptr, *d = nullptr;
        if (svt_ctx_stage(ctx, 8 * sizeof(void *), &h, &d)) return svt_set_error(SVT_HIP_ERR_NO_RESOURCES, "tq: descriptor buffers");
        for (int i = 0; i < 8; i++) ((uint8_t **)h)[i] = i < n_set ? recon_set[i] : nullptr; /* a block that names a set >= n_set is not reconstructed anywhere */
        HIP_TRY(hipMemcpyAsync(d, h, 8 * sizeof(void *), hipMemcpyHostToDevice, ctx->stream));
        d_set = (const uint8_t *const *)d;
    }
    HIP_TRY(hipEventRecord(ctx->ev_start, ctx->stream));
    /* The four size groups are independent of each other; SVT_HIP_TQ_FORK=1 launches each on a stream of its own (forked from
     * and joined into the context's stream).  Measured: slower -- 1.59 instead of 1.39 ms per mini-GOP for the four launches
     * alone, no change inside the pipeline: kernels that share compute units take longer in sum than one after the other (every
     * pairing of the stages shows it), so the default keeps them in sequence. */
    static const bool want_fork = getenv("SVT_HIP_TQ_FORK") != nullptr;
    const bool        fork = want_fork && svt_ctx_aux_init(ctx) == 0;
    if (fork) {
        HIP_TRY(hipEventRecord(ctx->aux_fork, ctx->stream));
        for (int i = 0; i < 3; i++) HIP_TRY(hipStreamWaitEvent(ctx->aux[i], ctx->aux_fork, 0));
    }
    int        off = 0;
    hipError_t rc = hipSuccess;
    auto at = [&](int o) { tq_rate_args r = ra; if (r.bits) r.bits += o; return r; };
    auto on = [&](int i) { return fork && i > 0 ? ctx->aux[i - 1] : ctx->stream; };
    rc = launch_tq<4, RATE>(ctx, on(0), d_src, d_pred, d_recon, d_blocks + off, size_count[0], d_qtabs, d_iscan, d_qcoeff, d_dqcoeff, d_eob + off, d_dist ? d_dist + 2 * off : nullptr, at(off), d_set);
    off += size_count[0];
    if (rc == hipSuccess) rc = launch_tq<8, RATE>(ctx, on(1), d_src, d_pred, d_recon, d_blocks + off, size_count[1], d_qtabs, d_iscan, d_qcoeff, d_dqcoeff, d_eob + off, d_dist ? d_dist + 2 * off : nullptr, at(off), d_set);
    off += size_count[1];
    if (rc == hipSuccess) rc = launch_tq<16, RATE>(ctx, on(2), d_src, d_pred, d_recon, d_blocks + off, size_count[2], d_qtabs, d_iscan, d_qcoeff, d_dqcoeff, d_eob + off, d_dist ? d_dist + 2 * off : nullptr, at(off), d_set);
    off += size_count[2];
    if (rc == hipSuccess) rc = launch_tq<32, RATE>(ctx, on(3), d_src, d_pred, d_recon, d_blocks + off, size_count[3], d_qtabs, d_iscan, d_qcoeff, d_dqcoeff, d_eob + off, d_dist ? d_dist + 2 * off : nullptr, at(off), d_set);
    if (fork)
        for (int i = 0; i < 3; i++) {
            (void)hipEventRecord(ctx->aux_join[i], ctx->aux[i]);
            (void)hipStreamWaitEvent(ctx->stream, ctx->aux_join[i], 0);
        }
    (void)hipEventRecord(ctx->ev_stop, ctx->stream); /* also on a failed launch: ev_start is already in the stream */
    if (recon_set) svt_ctx_stage_commit(ctx);
    if (rc != hipSuccess) return svt_set_hip_error(rc, __FILE__, __LINE__);
    ctx->timed = 1;
    return SVT_HIP_OK;
}
} // namespace

/* Transform stage over block lists that were built ON THE DEVICE (csrc/encdec.hip): d_off_cnt[s] / d_off_cnt[4 + s] = first block and
 * number of blocks of size s inside d_blocks / d_eob; cap[s] = an upper bound known to the host, which only sizes the persistent
 * grids.  No rate, optional distortion.  d_pos != null: the lists are position codes (d_blocks is not read): tq_dev_count.  Internal to the
 * library (declared in svt_ctx.h). */
int32_t svt_tq_launch_device_lists(svt_hip_ctx *ctx, const uint8_t *d_src, const uint8_t *d_pred, uint8_t *const *recon_set, int n_set,
                                   const svt_tq_block *d_blocks, const int32_t cap[4], const int32_t *d_off_cnt, const svt_quant_tables *d_qtabs,
                                   const int16_t *d_iscan, int16_t *d_qcoeff, int16_t *d_dqcoeff, uint16_t *d_eob, uint64_t *d_dist,
                                   const uint32_t *d_pos, const void *d_geom, int geom_stride, const uint32_t *d_iscan_off, int sb_cols) {
    HIP_TRY(hipSetDevice(ctx->device));
    void *h = nullptr, *d = nullptr;
    if (svt_ctx_stage(ctx, 8 * sizeof(void *), &h, &d)) return svt_set_error(SVT_HIP_ERR_NO_RESOURCES, "tq: descriptor buffers");
    for (int i = 0; i < 8; i++) ((uint8_t **)h)[i] = i < n_set ? recon_set[i] : nullptr; /* a block that names a set >= n_set is not reconstructed anywhere */
    HIP_TRY(hipMemcpyAsync(d, h, 8 * sizeof(void *), hipMemcpyHostToDevice, ctx->stream));
    const uint8_t *const *d_set = (const uint8_t *const *)d;
    const tq_rate_args none = {nullptr, nullptr, nullptr};
    hipError_t rc = hipSuccess;
#define TQ_DEV(N, S, D) launch_tq<N, false, D>(ctx, ctx->stream, d_src, d_pred, nullptr, d_blocks, cap[S], d_qtabs, d_iscan, d_qcoeff, d_dqcoeff, d_eob, d_dist, none, d_set, tq_dev_count{d_off_cnt, S, d_pos, (const uint8_t *)d_geom, geom_stride, d_iscan_off, sb_cols})
    if (d_dist) { /* with the coefficient-domain distortion pair */
        rc = TQ_DEV(4, 0, true);
        if (rc == hipSuccess) rc = TQ_DEV(8, 1, true);
        if (rc == hipSuccess) rc = TQ_DEV(16, 2, true);
        if (rc == hipSuccess) rc = TQ_DEV(32, 3, true);
    } else {      /* the encode pass: no distortion sums (instances without that arithmetic) */
        rc = TQ_DEV(4, 0, false);
        if (rc == hipSuccess) rc = TQ_DEV(8, 1, false);
        if (rc == hipSuccess) rc = TQ_DEV(16, 2, false);
        if (rc == hipSuccess) rc = TQ_DEV(32, 3, false);
    }
#undef TQ_DEV
    svt_ctx_stage_commit(ctx);
    if (rc != hipSuccess) return svt_set_hip_error(rc, __FILE__, __LINE__);
    return SVT_HIP_OK;
}

/* The same stage over the SB-ordered device lists (csrc/encdec.hip): ONE launch, workgroup = (picture, chunk, size, part).  d_seg: per picture
 * n_chunks x seg_per_chunk exclusive prefixes, d_bases: n_pics + 1 first blocks.  No distortion, no rate (the encode pass).  Internal. */
int32_t svt_tq_launch_sb_lists(svt_hip_ctx *ctx, const uint8_t *d_src, const uint8_t *d_pred, uint8_t *const *recon_set, int n_set, const svt_quant_tables *d_qtabs,
                               const int16_t *d_iscan, int16_t *d_qcoeff, int16_t *d_dqcoeff, uint16_t *d_eob, const uint32_t *d_pos, const void *d_geom, int geom_stride,
                               const uint32_t *d_iscan_off, int sb_cols, const int32_t *d_seg, const int32_t *d_bases, int n_pics, int n_chunks, int seg_per_chunk) {
    HIP_TRY(hipSetDevice(ctx->device));
    void *h = nullptr, *d = nullptr;
    if (svt_ctx_stage(ctx, 8 * sizeof(void *), &h, &d)) return svt_set_error(SVT_HIP_ERR_NO_RESOURCES, "tq: descriptor buffers");
    for (int i = 0; i < 8; i++) ((uint8_t **)h)[i] = i < n_set ? recon_set[i] : nullptr;
    HIP_TRY(hipMemcpyAsync(d, h, 8 * sizeof(void *), hipMemcpyHostToDevice, ctx->stream));
    const int n_items = n_pics * n_chunks, groups8 = (n_items + 7) / 8;
    const tq_dev_count dc{nullptr, 0, d_pos, (const uint8_t *)d_geom, geom_stride, d_iscan_off, sb_cols};
    /* SVT_HIP_TQ_SB_SPLIT: 0 (default) = everything in one launch; 1 = sizes 4x4 .. 16x16 in one launch (96 registers, 8.7 KB of LDS), the 32x32
     * blocks in a second; 2 = 4x4 + 8x8 | 16x16 + 32x32.  Measured (tools/r06_tq_ab.sh, tools/r06_tq_traffic.sh): traffic 1.50 / 1.73 GB per 16
     * pictures for 0 / 1, step 10.5 / 10.4 / 10.65 ms for 0 / 1 / 2 against 10.2 ms with the size-grouped lists. */
    static const int split = getenv("SVT_HIP_TQ_SB_SPLIT") ? atoi(getenv("SVT_HIP_TQ_SB_SPLIT")) : 0;
#define TQ_SB_LAUNCH(LO, HI) hipLaunchKernelGGL((svt_tq_sb_kernel<LO, HI>), dim3(groups8 * tq_sb_slots_range(LO, HI) * 8), dim3(TQ_SB_NT), 0, ctx->stream, d_src, d_pred, d_qtabs, d_iscan, \
                                                d_qcoeff, d_dqcoeff, d_eob, (const uint8_t *const *)d, dc, d_seg, d_bases, n_chunks, seg_per_chunk, n_items)
    if (split == 0) TQ_SB_LAUNCH(0, 3);
    else if (split == 2) { TQ_SB_LAUNCH(0, 1); TQ_SB_LAUNCH(2, 3); }
    else { TQ_SB_LAUNCH(0, 2); TQ_SB_LAUNCH(3, 3); }
#undef TQ_SB_LAUNCH
    const hipError_t rc = hipGetLastError();
    svt_ctx_stage_commit(ctx);
    if (rc != hipSuccess) return svt_set_hip_error(rc, __FILE__, __LINE__);
    return SVT_HIP_OK;
}

/* Blocks must be grouped by transform size: size_count[s] blocks of SVT_TX_<s>, in the order 4x4, 8x8, 16x16,
 * 32x32, and within a size all blocks must share the same do_recon flag (the encode pass reconstructs every
 * block; mode decision none) -- the kernel keeps its barriers workgroup-uniform that way. */
extern "C" int32_t svt_hip_tq_batch_device(svt_hip_ctx *ctx, const uint8_t *d_src, const uint8_t *d_pred, uint8_t *d_recon,
                                           const svt_tq_block *d_blocks, const int32_t size_count[4],
                                           const svt_quant_tables *d_qtabs, const int16_t *d_iscan, int16_t *d_qcoeff,
                                           int16_t *d_dqcoeff, uint16_t *d_eob) {
    return svt_hip_tq_batch_dist_device(ctx, d_src, d_pred, d_recon, d_blocks, size_count, d_qtabs, d_iscan, d_qcoeff, d_dqcoeff, d_eob,
                                        nullptr);
}

/* as svt_hip_tq_batch_device, plus d_dist[2*b], d_dist[2*b+1] = full_distortion_kernel32bit's result pair of block b
 * (residual distortion, prediction distortion); d_dist may be NULL */
extern "C" int32_t svt_hip_tq_batch_dist_device(svt_hip_ctx *ctx, const uint8_t *d_src, const uint8_t *d_pred, uint8_t *d_recon,
                                                const svt_tq_block *d_blocks, const int32_t size_count[4],
                                                const svt_quant_tables *d_qtabs, const int16_t *d_iscan, int16_t *d_qcoeff,
                                                int16_t *d_dqcoeff, uint16_t *d_eob, uint64_t *d_dist) {
    if (!ctx || !d_src || !d_pred || !d_blocks || !size_count || !d_qtabs || !d_iscan || !d_qcoeff || !d_dqcoeff || !d_eob)
        return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "tq: null argument");
    if (((uintptr_t)d_qcoeff | (uintptr_t)d_dqcoeff) & 15) return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "tq: coefficient arrays must be 16-byte aligned");
    tq_rate_args none = {nullptr, nullptr, nullptr};
    return tq_launch_all<false>(ctx, d_src, d_pred, d_recon, d_blocks, size_count, d_qtabs, d_iscan, d_qcoeff, d_dqcoeff, d_eob, d_dist, none);
}

/* as svt_hip_tq_batch_dist_device, plus d_bits[b] = coeff_rate_estimate of block b computed behind the quantiser (no second
 * pass over the coefficients): the rate inputs of a block travel in svt_tq_block.pad_[0] (SVT_TQ_RATE_INFO) */
extern "C" int32_t svt_hip_tq_rd_batch_device(svt_hip_ctx *ctx, const uint8_t *d_src, const uint8_t *d_pred, uint8_t *d_recon,
                                              const svt_tq_block *d_blocks, const int32_t size_count[4],
                                              const svt_quant_tables *d_qtabs, const int16_t *d_iscan, int16_t *d_qcoeff,
                                              int16_t *d_dqcoeff, uint16_t *d_eob, uint64_t *d_dist,
                                              const svt_rate_tables *d_tables, const int16_t *d_scan, int32_t *d_bits) {
    if (!ctx || !d_src || !d_pred || !d_blocks || !size_count || !d_qtabs || !d_iscan || !d_qcoeff || !d_dqcoeff || !d_eob || !d_tables ||
        !d_scan || !d_bits)
        return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "tq_rd: null argument");
    if (((uintptr_t)d_qcoeff | (uintptr_t)d_dqcoeff) & 15) return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "tq_rd: coefficient arrays must be 16-byte aligned");
    if ((uintptr_t)d_scan & 3) return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "tq_rd: scan array must be 4-byte aligned");
    tq_rate_args ra = {d_tables, d_scan, d_bits};
    return tq_launch_all<true>(ctx, d_src, d_pred, d_recon, d_blocks, size_count, d_qtabs, d_iscan, d_qcoeff, d_dqcoeff, d_eob, d_dist, ra);
}

/* svt_hip_tq_rd_batch_device for a batch whose blocks reconstruct into up to 8 different buffers: block b writes into
 * recon_set[(pad_[0] >> 4) & 7] + recon_off.  One launch per transform size then serves pictures whose reference buffers lie further
 * apart than a 32-bit offset reaches, or in separate allocations -- e.g. the pictures of five consecutive mini-GOPs that one step of
 * a picture-level pipeline codes (bench.py's diagonal schedule). */
extern "C" int32_t svt_hip_tq_rd_batch_multi_device(svt_hip_ctx *ctx, const uint8_t *d_src, const uint8_t *d_pred, uint8_t *const *recon_set, int32_t n_set,
                                                    const svt_tq_block *d_blocks, const int32_t size_count[4], const svt_quant_tables *d_qtabs,
                                                    const int16_t *d_iscan, int16_t *d_qcoeff, int16_t *d_dqcoeff, uint16_t *d_eob, uint64_t *d_dist,
                                                    const svt_rate_tables *d_tables, const int16_t *d_scan, int32_t *d_bits) {
    if (!ctx || !d_src || !d_pred || !recon_set || n_set < 1 || n_set > 8 || !d_blocks || !size_count || !d_qtabs || !d_iscan || !d_qcoeff || !d_dqcoeff || !d_eob)
        return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "tq_multi: bad argument");
    if (((uintptr_t)d_qcoeff | (uintptr_t)d_dqcoeff) & 15) return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "tq_multi: coefficient arrays must be 16-byte aligned");
    if (d_bits) {
        if (!d_tables || !d_scan || ((uintptr_t)d_scan & 3)) return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "tq_multi: rate tables / scan array");
        tq_rate_args ra = {d_tables, d_scan, d_bits};
        return tq_launch_all<true>(ctx, d_src, d_pred, nullptr, d_blocks, size_count, d_qtabs, d_iscan, d_qcoeff, d_dqcoeff, d_eob, d_dist, ra, recon_set, n_set);
    }
    tq_rate_args none = {nullptr, nullptr, nullptr};
    return tq_launch_all<false>(ctx, d_src, d_pred, nullptr, d_blocks, size_count, d_qtabs, d_iscan, d_qcoeff, d_dqcoeff, d_eob, d_dist, none, recon_set, n_set);
}

/* the compiled-in 4x4 scan orders of the in-lane rate pass: out[0..15] = scan, out[16..47] = the neighbour pairs of positions 0..15 */
extern "C" int32_t svt_hip_rate_scan4x4_table(int32_t tx_type, int16_t out[48]) {
    if (tx_type < 0 || tx_type > 3 || !out) return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "scan4x4: bad argument");
    const scan4_tab &t = SCAN4[scan4_kind(tx_type)];
    for (int i = 0; i < 16; i++) out[i] = t.scan[i];
    for (int i = 0; i < 32; i++) out[16 + i] = t.nb[i];
    return SVT_HIP_OK;
}

extern "C" int32_t svt_hip_tq_batch(svt_hip_ctx *ctx, const uint8_t *src, const uint8_t *pred, uint8_t *recon, size_t plane_bytes,
                                    const svt_tq_block *blocks, int32_t n_blocks, const svt_quant_tables *qtabs, int32_t n_qtabs,
                                    const int16_t *iscan, size_t iscan_count, int16_t *qcoeff, int16_t *dqcoeff,
                                    size_t coeff_count, uint16_t *eob) {
    if (!ctx || !src || !pred || !recon || !blocks || n_blocks < 1 || !qtabs || !iscan || !qcoeff || !dqcoeff || !eob)
        return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "tq: null argument");
    int32_t cnt[4] = {0, 0, 0, 0};
    for (int i = 0; i < n_blocks; i++) {
        if (blocks[i].tx_size > 3 || blocks[i].qtab >= n_qtabs) return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "tq: bad block");
        if (blocks[i].coeff_off & 7) return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "tq: coeff_off must be a multiple of 8 coefficients");
        if (i && blocks[i].tx_size < blocks[i - 1].tx_size) return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "tq: blocks not grouped by tx_size");
        if (i && blocks[i].tx_size == blocks[i - 1].tx_size && (blocks[i].do_recon != 0) != (blocks[i - 1].do_recon != 0))
            return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "tq: do_recon must be uniform within a tx_size group");
        cnt[blocks[i].tx_size]++;
    }
    HIP_TRY(hipSetDevice(ctx->device));
    uint8_t *ds = (uint8_t *)svt_ctx_slot(ctx, 11, plane_bytes + 64), *dp = (uint8_t *)svt_ctx_slot(ctx, 12, plane_bytes + 64);
    uint8_t *dr = (uint8_t *)svt_ctx_slot(ctx, 13, plane_bytes + 64);
    svt_tq_block *db = (svt_tq_block *)svt_ctx_slot(ctx, 14, sizeof(svt_tq_block) * (size_t)n_blocks);
    svt_quant_tables *dqt = (svt_quant_tables *)svt_ctx_slot(ctx, 15, sizeof(svt_quant_tables) * (size_t)n_qtabs);
    int16_t *dis = (int16_t *)svt_ctx_slot(ctx, 16, sizeof(int16_t) * iscan_count);
    int16_t *dq = (int16_t *)svt_ctx_slot(ctx, 17, sizeof(int16_t) * coeff_count), *ddq = (int16_t *)svt_ctx_slot(ctx, 18, sizeof(int16_t) * coeff_count);
    uint16_t *de = (uint16_t *)svt_ctx_slot(ctx, 19, sizeof(uint16_t) * (size_t)n_blocks);
    if (!ds || !dp || !dr || !db || !dqt || !dis || !dq || !ddq || !de) return svt_set_error(SVT_HIP_ERR_NO_RESOURCES, "tq: device buffers");
    HIP_TRY(hipMemcpyAsync(ds, src, plane_bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(dp, pred, plane_bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(dr, recon, plane_bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(db, blocks, sizeof(svt_tq_block) * (size_t)n_blocks, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(dqt, qtabs, sizeof(svt_quant_tables) * (size_t)n_qtabs, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(dis, iscan, sizeof(int16_t) * iscan_count, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemsetAsync(dq, 0, sizeof(int16_t) * coeff_count, ctx->stream));
    HIP_TRY(hipMemsetAsync(ddq, 0, sizeof(int16_t) * coeff_count, ctx->stream));
    int32_t rc = svt_hip_tq_batch_device(ctx, ds, dp, dr, db, cnt, dqt, dis, dq, ddq, de);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(recon, dr, plane_bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(qcoeff, dq, sizeof(int16_t) * coeff_count, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(dqcoeff, ddq, sizeof(int16_t) * coeff_count, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(eob, de, sizeof(uint16_t) * (size_t)n_blocks, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return SVT_HIP_OK;
}
