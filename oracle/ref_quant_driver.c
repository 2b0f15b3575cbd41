/*
 * ref_quant_driver.c -- harness that runs the REFERENCE's eb_vp9_init_quantizer (Source/Lib/VPX/vp9_quantize.c:206) and
 * prints, for every q index, the step sizes and the tables it derives.  TEST INFRASTRUCTURE ONLY (rules: ref_me_driver.c).
 * argv: y_dc_delta_q uv_dc_delta_q uv_ac_delta_q.  Output lines:
 *   q y_dc_step(delta 0) | Y: dc ac zbin[2] round[2] quant[2] quant_shift[2] dequant[2] | UV: the same
 */
#include <stdio.h>
#include <stdlib.h>
#include "vp9_encoder.h"
#include "vp9_quantize.h"
#include "vp9_quant_common.h"

uint32_t eb_vp9_ASM_TYPES = 0;

int main(int argc, char **argv) {
    VP9_COMP *cpi = (VP9_COMP *)calloc(1, sizeof *cpi);
    cpi->common.bit_depth = VPX_BITS_8;
    if (argc > 3) { cpi->common.y_dc_delta_q = atoi(argv[1]); cpi->common.uv_dc_delta_q = atoi(argv[2]); cpi->common.uv_ac_delta_q = atoi(argv[3]); }
    eb_vp9_init_quantizer(cpi);
    const QUANTS *Q = &cpi->quants;
    for (int q = 0; q < QINDEX_RANGE; q++) {
        printf("%d %d", q, eb_vp9_dc_quant(q, 0, VPX_BITS_8));
        printf(" %d %d", eb_vp9_dc_quant(q, cpi->common.y_dc_delta_q, VPX_BITS_8), eb_vp9_ac_quant(q, 0, VPX_BITS_8));
        for (int i = 0; i < 2; i++) printf(" %d %d %d %d %d", Q->y_zbin[q][i], Q->y_round[q][i], Q->y_quant[q][i], Q->y_quant_shift[q][i], cpi->y_dequant[q][i]);
        printf(" %d %d", eb_vp9_dc_quant(q, cpi->common.uv_dc_delta_q, VPX_BITS_8), eb_vp9_ac_quant(q, cpi->common.uv_ac_delta_q, VPX_BITS_8));
        for (int i = 0; i < 2; i++) printf(" %d %d %d %d %d", Q->uv_zbin[q][i], Q->uv_round[q][i], Q->uv_quant[q][i], Q->uv_quant_shift[q][i], cpi->uv_dequant[q][i]);
        printf("\n");
    }
    return 0;
}
