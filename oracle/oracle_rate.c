/*
 * oracle_rate.c -- CPU restatement (TEST INFRASTRUCTURE ONLY) of the reference's coefficient rate estimation.
 *   coeff_rate_estimate (= libvpx cost_coeffs)   Source/Lib/Codec/EbRateDistortionCost.c:55-172, non-fast branch :131-169
 *   band_counts                                   Source/Lib/Codec/EbRateDistortionCost.c:47-52
 *   vp9_get_token_cost                            Source/Lib/VPX/vp9_tokenize.h:118-127
 *   value -> token (dct_cat_lt_10_value_tokens)   Source/Lib/VPX/vp9_tokenize.c:36-50: ZERO, ONE..FOUR, CAT1 5-6, CAT2 7-10,
 *                                                 CAT3 11-18, CAT4 19-34, CAT5 35-66, CAT6 >= 67 (VPX/vp9_entropy.h:28-52)
 *   eb_vp9_pt_energy_class                        Source/Lib/VPX/vp9_entropy.c:111
 *   get_coef_context                              Source/Lib/VPX/vp9_entropy.h:184-189
 * Pinned against the reference's own coeff_rate_estimate() built from source (oracle/_ref/ref_rate_blocks) on the tables
 * its own fill_token_costs produces from the default coefficient probabilities; tests/golden/rate_reference.npz.
 */
#include <stdlib.h>
#include "oracle.h"

static int token_of(int v) {
    const int a = v < 0 ? -v : v;
    return a < 5 ? a : a < 7 ? 5 : a < 11 ? 6 : a < 19 ? 7 : a < 35 ? 8 : a < 67 ? 9 : 10;
}
static const uint8_t k_energy[12] = {0, 1, 2, 3, 3, 4, 4, 5, 5, 5, 5, 5};
static const int16_t k_band_counts[4][8] = {{1, 2, 3, 4, 3, 16 - 13, 0}, {1, 2, 3, 4, 11, 64 - 21, 0}, {1, 2, 3, 4, 11, 256 - 21, 0}, {1, 2, 3, 4, 11, 1024 - 21, 0}};

static int value_cost(const svt_rate_tables *t, int v, int *tok) {
    *tok = token_of(v);
    if (*tok == 10) {
        const int extra = abs(v) - 67;
        return t->cat6_low_cost[extra & 0xff] + t->cat6_high_cost[extra >> 8];
    }
    return t->value_cost[v + 66];
}

int32_t svt_oracle_coeff_rate_batch(const int16_t *qcoeff, const svt_rate_block *blocks, int32_t n_blocks, const svt_rate_tables *t,
                                    const int16_t *scan_all, int32_t *bits) {
    for (int b = 0; b < n_blocks; b++) {
        const svt_rate_block *k = &blocks[b];
        const int             n = 16 << (2 * k->tx_size);
        const int16_t        *q = qcoeff + k->coeff_off, *scan = scan_all + k->scan_off, *nb = scan + n;
        const uint32_t (*tc)[2][6][12] = t->token_costs[k->tx_size][k->plane_type][k->is_inter]; /* [band][!prev][ctx][token] */
        uint8_t  cache[32 * 32];
        int      cost;
        if (k->eob == 0) {
            cost = (int)tc[0][0][k->ctx][11];
        } else {
            const int16_t *band_count = &k_band_counts[k->tx_size][1];
            int            band_left = *band_count++, band = 0, tok, c, pt;
            cost = value_cost(t, q[0], &tok);
            cost += (int)tc[0][0][k->ctx][tok];
            cache[0] = k_energy[tok];
            ++band;
            int prev_zero = !tok;
            for (c = 1; c < k->eob; c++) {
                const int rc = scan[c];
                cost += value_cost(t, q[rc], &tok);
                pt = (1 + cache[nb[2 * c]] + cache[nb[2 * c + 1]]) >> 1;
                cost += (int)tc[band][prev_zero][pt][tok];
                cache[rc] = k_energy[tok];
                if (!--band_left) { band_left = *band_count++; ++band; }
                prev_zero = !tok;
            }
            if (band_left) {
                pt = (1 + cache[nb[2 * c]] + cache[nb[2 * c + 1]]) >> 1;
                cost += (int)tc[band][0][pt][11];
            }
        }
        bits[b] = cost;
    }
    return 0;
}
