/*
 * oracle_tq.c -- CPU restatement of the transform / quantisation path of perform_coding_loop
 * (Source/Lib/Codec/EbEncDecProcess.c:365-587): residual -> forward DCT/ADST -> quantise (+dequantise)
 * -> inverse transform added onto the prediction.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle_me.c for the rules).  Pinned bit-exact against the reference's own
 * C kernels (oracle/_ref/libsvtref_kernels.so) in tests/test_oracle_vs_ref.py.
 * Paths below are relative to /root/reference/Source/Lib.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/svtvp9_hip.h"
#include "oracle.h"
#include "oracle_txfm1d.h"

typedef void (*tx1d_fn)(const int32_t *, int32_t *);
static void fdct8_c16(const int32_t *in, int32_t *out) { tx_fdct8(in, out, 1); }
static void fdct4_w(const int32_t *in, int32_t *out) { tx_fdct4(in, out); for (int i = 0; i < 4; i++) out[i] = (int16_t)out[i]; }
static void fdct8_w(const int32_t *in, int32_t *out) { fdct8_c16(in, out); for (int i = 0; i < 8; i++) out[i] = (int16_t)out[i]; }
static void fdct16_w(const int32_t *in, int32_t *out) { tx_fdct16(in, out); for (int i = 0; i < 16; i++) out[i] = (int16_t)out[i]; }

/* ---- forward 2-D, DCT_DCT: VPX/fwd_txfm.c ---- */
/* eb_vp9_fdct4x4_c :15-79 */
static void fdct4x4(const int16_t *in, int stride, int16_t *out) {
    int16_t mid[16];
    for (int c = 0; c < 4; c++) {
        int32_t v[4], o[4];
        for (int r = 0; r < 4; r++) v[r] = in[r * stride + c] * 16;
        if (c == 0 && v[0]) ++v[0];
        tx_fdct4(v, o);
        for (int k = 0; k < 4; k++) mid[c * 4 + k] = (int16_t)o[k];
    }
    for (int i = 0; i < 4; i++) {
        int32_t v[4], o[4];
        for (int k = 0; k < 4; k++) v[k] = mid[k * 4 + i];
        tx_fdct4(v, o);
        for (int k = 0; k < 4; k++) out[i * 4 + k] = (int16_t)(((int16_t)o[k] + 1) >> 2);
    }
}
/* eb_vp9_fdct8x8_c :90-170 (final "/= 2" is C division, truncating toward zero) */
static void fdct8x8(const int16_t *in, int stride, int16_t *out) {
    int16_t mid[64];
    for (int c = 0; c < 8; c++) {
        int32_t v[8], o[8];
        for (int r = 0; r < 8; r++) v[r] = in[r * stride + c] * 4;
        tx_fdct8(v, o, 0);
        for (int k = 0; k < 8; k++) mid[c * 8 + k] = (int16_t)o[k];
    }
    for (int i = 0; i < 8; i++) {
        int32_t v[8], o[8];
        for (int k = 0; k < 8; k++) v[k] = mid[k * 8 + i];
        tx_fdct8(v, o, 0);
        for (int k = 0; k < 8; k++) out[i * 8 + k] = (int16_t)((int16_t)o[k] / 2);
    }
}
/* eb_vp9_fdct16x16_c :183-366 */
static void fdct16x16(const int16_t *in, int stride, int16_t *out) {
    int16_t mid[256];
    for (int c = 0; c < 16; c++) {
        int32_t v[16], o[16];
        for (int r = 0; r < 16; r++) v[r] = in[r * stride + c] * 4;
        tx_fdct16(v, o);
        for (int k = 0; k < 16; k++) mid[c * 16 + k] = (int16_t)o[k];
    }
    for (int i = 0; i < 16; i++) {
        int32_t v[16], o[16];
        for (int k = 0; k < 16; k++) v[k] = (mid[k * 16 + i] + 1) >> 2;
        tx_fdct16(v, o);
        for (int k = 0; k < 16; k++) out[i * 16 + k] = (int16_t)o[k];
    }
}
/* eb_vp9_fdct32x32_c :708-726 and eb_vpx_partial_fdct32x32_c :1051-1071 (low 16x16 only; the caller
 * pre-zeros the output, EbEncDecProcess.c:404) */
static void fdct32x32(const int16_t *in, int stride, int16_t *out, int partial) {
    int32_t *mid = (int32_t *)malloc(sizeof(int32_t) * 1024);
    const int n = partial ? 16 : 32;
    if (partial) memset(out, 0, sizeof(int16_t) * 1024);
    memset(mid, 0, sizeof(int32_t) * 1024);
    for (int c = 0; c < 32; c++) {
        int32_t v[32], o[32];
        for (int r = 0; r < 32; r++) v[r] = in[r * stride + c] * 4;
        tx_fdct32(v, o);
        for (int k = 0; k < n; k++) mid[k * 32 + c] = (o[k] + 1 + (o[k] > 0)) >> 2;
    }
    for (int i = 0; i < n; i++) {
        int32_t o[32];
        tx_fdct32(&mid[i * 32], o);
        for (int k = 0; k < n; k++) out[i * 32 + k] = (int16_t)((o[k] + 1 + (o[k] < 0)) >> 2);
    }
    free(mid);
}

/* ---- forward 2-D hybrid: VPX/vp9_dct.c eb_vp9_fht4x4_c :516-540, fht8x8_c :653-675, fht16x16_c :734-756 ---- */
static void fht(const int16_t *in, int stride, int16_t *out, int n, int tx_type) {
    /* tx_type: 1 ADST_DCT (cols ADST, rows DCT), 2 DCT_ADST, 3 ADST_ADST (transform_2d = {cols, rows}) */
    tx1d_fn dct = n == 4 ? fdct4_w : n == 8 ? fdct8_w : fdct16_w;
    tx1d_fn adst = n == 4 ? tx_fadst4 : n == 8 ? tx_adst8 : tx_adst16;
    tx1d_fn cols = (tx_type == SVT_ADST_DCT || tx_type == SVT_ADST_ADST) ? adst : dct;
    tx1d_fn rows = (tx_type == SVT_DCT_ADST || tx_type == SVT_ADST_ADST) ? adst : dct;
    int16_t mid[256];
    for (int c = 0; c < n; c++) {
        int32_t v[16], o[16];
        for (int r = 0; r < n; r++) v[r] = (int16_t)(in[r * stride + c] * (n == 4 ? 16 : 4));
        if (n == 4 && c == 0 && v[0]) v[0] = (int16_t)(v[0] + 1);
        cols(v, o);
        for (int k = 0; k < n; k++) mid[k * n + c] = n == 16 ? (int16_t)((o[k] + 1 + (o[k] < 0)) >> 2) : (int16_t)o[k];
    }
    for (int i = 0; i < n; i++) {
        int32_t v[16], o[16];
        for (int k = 0; k < n; k++) v[k] = mid[i * n + k];
        rows(v, o);
        for (int k = 0; k < n; k++) {
            int32_t t = o[k];
            out[i * n + k] = n == 4 ? (int16_t)((t + 1) >> 2) : n == 8 ? (int16_t)((t + (t < 0)) >> 1) : (int16_t)t;
        }
    }
}

void oracle_fwd_txfm(const int16_t *residual, int stride, int16_t *coeff, int tx_size, int tx_type, int partial32) {
    switch (tx_size) {
    case SVT_TX_4X4: if (tx_type == SVT_DCT_DCT) fdct4x4(residual, stride, coeff); else fht(residual, stride, coeff, 4, tx_type); break;
    case SVT_TX_8X8: if (tx_type == SVT_DCT_DCT) fdct8x8(residual, stride, coeff); else fht(residual, stride, coeff, 8, tx_type); break;
    case SVT_TX_16X16: if (tx_type == SVT_DCT_DCT) fdct16x16(residual, stride, coeff); else fht(residual, stride, coeff, 16, tx_type); break;
    default: fdct32x32(residual, stride, coeff, partial32); break;
    }
}

/* ---- quantiser: VPX/quantize.c eb_vp9_quantize_b_c :112-157, eb_vp9_quantize_b_32x32_c :206-254 ----
 * The reference's "pre-scan" only skips coefficients that the per-coefficient zbin test would skip
 * anyway, so each coefficient is independent; eob = 1 + last scan position with a non-zero level. */
static int clamp16(int v) { return v < -32768 ? -32768 : v > 32767 ? 32767 : v; }
void oracle_quantize(const int16_t *coeff, int n, const svt_quant_tables *q, int16_t *qcoeff, int16_t *dqcoeff,
                     uint16_t *eob_out, const int16_t *iscan, int is32) {
    int eob = -1;
    for (int rc = 0; rc < n; rc++) {
        const int ac = rc != 0, c = coeff[rc], sign = c >> 31;
        int       level = 0;
        qcoeff[rc] = 0; dqcoeff[rc] = 0;
        if (!is32) {
            const int zbin = q->zbin[ac], a = (c ^ sign) - sign;
            if (!(c < zbin && c > -zbin) && a >= zbin) {
                int tmp = clamp16(a + q->round[ac]);
                level   = ((((tmp * q->quant[ac]) >> 16) + tmp) * q->quant_shift[ac]) >> 16;
                qcoeff[rc]  = (int16_t)((level ^ sign) - sign);
                dqcoeff[rc] = (int16_t)(qcoeff[rc] * q->dequant[ac]);
            }
        } else {
            const int zbin = (q->zbin[ac] + 1) >> 1;
            if (c >= zbin || c <= -zbin) {
                int a = (c ^ sign) - sign;
                a += (q->round[ac] + 1) >> 1;
                a = clamp16(a);
                level = ((((a * q->quant[ac]) >> 16) + a) * q->quant_shift[ac]) >> 15;
                qcoeff[rc]  = (int16_t)((level ^ sign) - sign);
                dqcoeff[rc] = (int16_t)(qcoeff[rc] * q->dequant[ac] / 2);
            }
        }
        if (level && iscan[rc] > eob) eob = iscan[rc];
    }
    *eob_out = (uint16_t)(int16_t)(eob + 1);
}

/* ---- inverse 2-D + add: VPX/inv_txfm.c *_add_c and VPX/vp9_idct.c wrappers :111-189 ----
 * rows first (row i of the coefficient block), then columns; `nrows` = how many coefficient rows the
 * eob-selected variant transforms (the others are taken as zero). */
static uint8_t clip_add(uint8_t d, int32_t t) { int v = d + t; return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }
static void inv2d_add(const int16_t *in, uint8_t *dst, int stride, int n, tx1d_fn rows, tx1d_fn cols, int nrows, int shift) {
    int16_t *mid = (int16_t *)calloc((size_t)n * n, sizeof(int16_t));
    for (int i = 0; i < nrows; i++) {
        int32_t v[32], o[32];
        for (int k = 0; k < n; k++) v[k] = in[i * n + k];
        rows(v, o);
        for (int k = 0; k < n; k++) mid[i * n + k] = (int16_t)o[k];
    }
    for (int c = 0; c < n; c++) {
        int32_t v[32], o[32];
        for (int r = 0; r < n; r++) v[r] = mid[r * n + c];
        cols(v, o);
        for (int r = 0; r < n; r++) {
            int32_t t = (int16_t)o[r];
            dst[r * stride + c] = clip_add(dst[r * stride + c], (t + (1 << (shift - 1))) >> shift);
        }
    }
    free(mid);
}
/* DC-only variants: eb_vp9_idct{4x4,8x8,16x16,32x32}_1_add_c (inv_txfm.c:174, 368, 784, 1241) */
static void inv_dc_add(const int16_t *in, uint8_t *dst, int stride, int n, int shift) {
    int32_t o = tx_rsw((int16_t)in[0] * TX_C16);
    o = tx_rsw(o * TX_C16);
    int32_t a1 = (o + (1 << (shift - 1))) >> shift;
    for (int r = 0; r < n; r++)
        for (int c = 0; c < n; c++) dst[r * stride + c] = clip_add(dst[r * stride + c], a1);
}

void oracle_inv_txfm_add(const int16_t *dqcoeff, uint8_t *dst, int stride, int tx_size, int tx_type, int eob) {
    const int n = 4 << tx_size;
    const int shift = tx_size == SVT_TX_4X4 ? 4 : tx_size == SVT_TX_8X8 ? 5 : 6;
    tx1d_fn   dct = n == 4 ? tx_idct4 : n == 8 ? tx_idct8 : n == 16 ? tx_idct16 : tx_idct32;
    if (tx_type == SVT_DCT_DCT || tx_size == SVT_TX_32X32) {
        int nrows = n;
        if (tx_size == SVT_TX_4X4) { if (eob <= 1) { inv_dc_add(dqcoeff, dst, stride, n, shift); return; } }
        else if (eob == 1) { inv_dc_add(dqcoeff, dst, stride, n, shift); return; }
        else if (tx_size == SVT_TX_8X8) nrows = eob <= 12 ? 4 : 8;
        else if (tx_size == SVT_TX_16X16) nrows = eob <= 10 ? 4 : eob <= 38 ? 8 : 16;
        else nrows = eob <= 34 ? 8 : eob <= 135 ? 16 : 32;
        inv2d_add(dqcoeff, dst, stride, n, dct, dct, nrows, shift);
        return;
    }
    tx1d_fn adst = n == 4 ? tx_iadst4 : n == 8 ? tx_adst8 : tx_adst16;
    tx1d_fn cols = (tx_type == SVT_ADST_DCT || tx_type == SVT_ADST_ADST) ? adst : dct;
    tx1d_fn rows = (tx_type == SVT_DCT_ADST || tx_type == SVT_ADST_ADST) ? adst : dct;
    inv2d_add(dqcoeff, dst, stride, n, rows, cols, n, shift);
}

int32_t svt_oracle_tq_batch_dist(const uint8_t *src, const uint8_t *pred, uint8_t *recon, const svt_tq_block *blocks,
                                 int32_t n_blocks, const svt_quant_tables *qtabs, const int16_t *iscan, int16_t *qcoeff,
                                 int16_t *dqcoeff, uint16_t *eob, uint64_t *dist);

/* T3: full_distortion_kernel32bit (C_DEFAULT/EbPictureOperators_C.c:288-311): out[0] = sum of squared differences
 * between the transform coefficients and the dequantised coefficients, out[1] = sum of squared coefficients.
 * [quirk] each difference is passed through an int16_t parameter (sq_r16to32, :264) and the sums are uint32_t. */
void svt_oracle_full_distortion32(const int16_t *coeff, const int16_t *recon_coeff, int32_t count, uint64_t out[2]) {
    uint32_t residual = 0, prediction = 0;
    for (int i = 0; i < count; i++) {
        const int16_t d = (int16_t)(coeff[i] - recon_coeff[i]);
        residual += (uint32_t)((int32_t)d * d);
        prediction += (uint32_t)((int32_t)coeff[i] * coeff[i]);
    }
    out[0] = residual;
    out[1] = prediction;
}

/* ---- batch driver: same contract as svt_hip_tq_batch (include/svtvp9_hip.h) ---- */
int32_t svt_oracle_tq_batch(const uint8_t *src, const uint8_t *pred, uint8_t *recon, const svt_tq_block *blocks,
                            int32_t n_blocks, const svt_quant_tables *qtabs, const int16_t *iscan, int16_t *qcoeff,
                            int16_t *dqcoeff, uint16_t *eob) {
    return svt_oracle_tq_batch_dist(src, pred, recon, blocks, n_blocks, qtabs, iscan, qcoeff, dqcoeff, eob, 0);
}

/* as above, plus the coefficient-domain distortion pair of every block (dist[2*b], dist[2*b+1]) when dist != 0 */
int32_t svt_oracle_tq_batch_dist(const uint8_t *src, const uint8_t *pred, uint8_t *recon, const svt_tq_block *blocks,
                                 int32_t n_blocks, const svt_quant_tables *qtabs, const int16_t *iscan, int16_t *qcoeff,
                                 int16_t *dqcoeff, uint16_t *eob, uint64_t *dist) {
    int16_t res[1024], coeff[1024];
    for (int b = 0; b < n_blocks; b++) {
        const svt_tq_block *k = &blocks[b];
        const int           n = 4 << k->tx_size;
        /* eb_vp9_residual_kernel, C_DEFAULT/EbPictureOperators_C.c:204-223 */
        for (int r = 0; r < n; r++)
            for (int c = 0; c < n; c++)
                res[r * n + c] = (int16_t)((int16_t)src[k->src_off + r * k->src_stride + c] - (int16_t)pred[k->pred_off + r * k->pred_stride + c]);
        oracle_fwd_txfm(res, n, coeff, k->tx_size, k->tx_size == SVT_TX_32X32 ? SVT_DCT_DCT : k->tx_type, k->partial32);
        oracle_quantize(coeff, n * n, &qtabs[k->qtab], qcoeff + k->coeff_off, dqcoeff + k->coeff_off, &eob[b], iscan + k->iscan_off,
                        k->tx_size == SVT_TX_32X32);
        if (dist) svt_oracle_full_distortion32(coeff, dqcoeff + k->coeff_off, n * n, &dist[2 * b]);
        if (k->do_recon) {
            /* pic_copy pred -> recon, then inverse transform added (EbEncDecProcess.c:430-437) */
            for (int r = 0; r < n; r++) memcpy(recon + k->recon_off + r * k->recon_stride, pred + k->pred_off + r * k->pred_stride, (size_t)n);
            if (eob[b])
                oracle_inv_txfm_add(dqcoeff + k->coeff_off, recon + k->recon_off, k->recon_stride, k->tx_size,
                                    k->tx_size == SVT_TX_32X32 ? SVT_DCT_DCT : k->tx_type, eob[b]);
        }
    }
    return 0;
}
