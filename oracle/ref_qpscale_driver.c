/*
 * ref_qpscale_driver.c -- harness around the REFERENCE's fixed-QP layer scaling: its own eb_vp9_compute_qdelta
 * (Source/Lib/VPX/vp9_ratectrl.c:2162), eb_vp9_convert_qindex_to_q (:158), eb_vp9_quantizer_to_qindex (VPX/vp9_quantize.c:329) and
 * its delta_rate_oq / delta_rate_sq / delta_rate_vmaf tables (the objects of Codec/EbRateControlProcess.c), combined exactly as
 * the QP_SCALING_MODE_0 branch of eb_vp9_rate_control_kernel does (Codec/EbRateControlProcess.c:4680-4722).  TEST INFRASTRUCTURE
 * ONLY (rules: ref_me_driver.c).  Output lines: tune hierarchical_levels layer qp base_qindex
 */
#include <stdio.h>
#include "vp9_encoder.h"
#include "vp9_ratectrl.h"
#include "vp9_quantize.h"
#include "EbDefinitions.h"

uint32_t eb_vp9_ASM_TYPES = 0;
extern const double delta_rate_oq[2][6];
extern const double delta_rate_sq[6];
extern const double delta_rate_vmaf[6];

int main(void) {
    RATE_CONTROL rc;
    rc.worst_quality = MAXQ;
    rc.best_quality  = MINQ;
    for (int tune = 0; tune < 3; tune++)
        for (int hl = 3; hl <= 4; hl++)
            for (int layer = 0; layer < 6; layer++)
                for (int qp = 0; qp < 64; qp++) {
                    const int    qindex = eb_vp9_quantizer_to_qindex(qp);
                    const double q = eb_vp9_convert_qindex_to_q(qindex, VPX_BITS_8);
                    const double r = tune == TUNE_OQ ? delta_rate_oq[hl == 4][layer] : tune == TUNE_SQ ? delta_rate_sq[layer] : delta_rate_vmaf[layer];
                    const int    d = eb_vp9_compute_qdelta(&rc, q, q * r, VPX_BITS_8);
                    printf("%d %d %d %d %d\n", tune, hl, layer, qp, VPXMAX(qindex + d, rc.best_quality));
                }
    return 0;
}
