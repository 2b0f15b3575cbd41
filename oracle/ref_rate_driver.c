/*
 * ref_rate_driver.c -- harness that runs the REFERENCE's coeff_rate_estimate() (Source/Lib/Codec/EbRateDistortionCost.c:55)
 * on a list of quantised transform blocks.  TEST INFRASTRUCTURE ONLY; compiled only in the build container against the
 * reference's headers and linked with the reference's own objects into oracle/_ref/ref_rate_blocks (rules: ref_me_driver.c).
 *
 * The token cost tables are produced by the reference itself: VPX/vp9_rd.c is compiled into this translation unit as it
 * lies (its fill_token_costs is static, :93-107) and run on the default coefficient probabilities
 * (eb_vp9_default_coef_probs, VPX/vp9_entropy.c:921), as vp9_initialize_rd_consts does for a key frame.
 *
 * request : int32 magic 'SVRT', n_blocks, then per block int32 {tx_size, plane, is_inter, tx_type, ctx, eob} + n*n int16
 * response: n_blocks int32 costs, then the tables the GPU path takes as input (written once, for the golden fixture):
 *           token_costs (13824 uint32), value_cost[-66..66] (133 int32), cat6_low_cost (256 int16), cat6_high_cost (64 uint16),
 *           and for tx_size 0..3 x tx_type 0..3: scan[n] + neighbors[2 (n + 1)] (int16)
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>

#define RTCD_C
#include "vpx_dsp_rtcd.h"
#include "vp9_rtcd.h"
#include "vp9_rd.c" /* the reference's file, for its static fill_token_costs */
#include "EbEncDecProcess.h"
#include "vp9_scan.h"
#include "vp9_tokenize.h"

uint32_t eb_vp9_ASM_TYPES = 0;

int coeff_rate_estimate(struct EncDecContext *context_ptr, MACROBLOCK *x, int16_t *trans_coeff_buffer, uint16_t eob, int plane, int block,
                        TX_SIZE tx_size, int pt, int use_fast_coef_costing);

static int rd(FILE *f, void *p, size_t n) { return fread(p, 1, n, f) == n ? 0 : -1; }

int main(int argc, char **argv) {
    if (argc < 3) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    int32_t h[2];
    if (rd(f, h, sizeof h) || h[0] != 0x54525653) return 3; /* 'SVRT' */
    const int n_blocks = h[1];

    VP9_COMMON *cm = (VP9_COMMON *)calloc(1, sizeof *cm);
    cm->fc = (FRAME_CONTEXT *)calloc(1, sizeof *cm->fc);
    eb_vp9_default_coef_probs(cm);
    MACROBLOCK *x = (MACROBLOCK *)calloc(1, sizeof *x);
    fill_token_costs(x->token_costs, cm->fc->coef_probs);

    EncDecContext *ctx = (EncDecContext *)calloc(1, sizeof *ctx);
    MACROBLOCKD   *xd  = (MACROBLOCKD *)calloc(1, sizeof *xd);
    ModeInfo       mi, *mip = &mi;
    ctx->e_mbd = xd; xd->mi = &mip;
    /* tx_type -> an intra mode that maps to it (eb_vp9_intra_mode_to_tx_type_lookup, VPX/vp9_blockd.c) */
    static const PREDICTION_MODE mode_of[4] = {DC_PRED, V_PRED, H_PRED, TM_PRED};

    int32_t *out = (int32_t *)malloc(sizeof(int32_t) * (size_t)n_blocks);
    int16_t  coef[32 * 32];
    for (int b = 0; b < n_blocks; b++) {
        int32_t d[6];
        if (rd(f, d, sizeof d)) return 3;
        const int n = 16 << (2 * d[0]);
        if (rd(f, coef, sizeof(int16_t) * (size_t)n)) return 3;
        memset(&mi, 0, sizeof mi);
        mi.sb_type      = BLOCK_64X64;
        mi.ref_frame[0] = d[2] ? LAST_FRAME : INTRA_FRAME;
        mi.ref_frame[1] = d[2] ? NONE : INTRA_FRAME;
        mi.mode         = d[2] ? NEWMV : mode_of[d[3] & 3];
        if (eb_vp9_intra_mode_to_tx_type_lookup[mode_of[d[3] & 3]] != (TX_TYPE)(d[3] & 3)) return 6;
        out[b] = coeff_rate_estimate(ctx, x, coef, (uint16_t)d[5], d[1], 0, (TX_SIZE)d[0], d[4], 0);
    }
    fclose(f);
    FILE *o = fopen(argv[2], "wb");
    if (!o) return 2;
    fwrite(out, sizeof(int32_t), (size_t)n_blocks, o);
    fwrite(x->token_costs, 1, sizeof x->token_costs, o);
    for (int v = -66; v <= 66; v++) { int32_t c = eb_vp9_dct_cat_lt_10_value_cost[v]; fwrite(&c, 4, 1, o); }
    fwrite(eb_vp9_cat6_low_cost, 2, 256, o);
    fwrite(eb_vp9_cat6_high_cost, 2, 64, o);
    for (int ts = 0; ts < 4; ts++)
        for (int tt = 0; tt < 4; tt++) {
            const scan_order *so = ts == 3 ? &eb_vp9_default_scan_orders[TX_32X32] : &eb_vp9_scan_orders[ts][tt];
            const int         n  = 16 << (2 * ts);
            fwrite(so->scan, 2, (size_t)n, o);
            fwrite(so->neighbors, 2, (size_t)(2 * (n + 1)), o);
        }
    fclose(o);
    return 0;
}
