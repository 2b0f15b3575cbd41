/*
 * qp_scaling.c -- the q index a picture is coded at in fixed-QP mode (rate_control_mode 0): the sequence QP scaled by the
 * picture's temporal layer, as the reference's rate-control kernel does for inter pictures (QP_SCALING_MODE_0,
 * Source/Lib/Codec/EbRateControlProcess.c:4680-4722; enable_qp_scaling_flag is always set, Codec/EbEncHandle.c:2034):
 *     qindex       = eb_vp9_quantizer_to_qindex(qp)
 *     q            = eb_vp9_convert_qindex_to_q(qindex)                       (= ac step / 4 at 8 bit, VPX/vp9_ratectrl.c:158)
 *     delta_qindex = eb_vp9_compute_qdelta(q, q * delta_rate[tune][layer])    (VPX/vp9_ratectrl.c:2162: first index whose q reaches
 *                                                                              the target minus first index whose q reaches q)
 *     base_qindex  = max(qindex + delta_qindex, MINQ)
 * with delta_rate_oq[hierarchical_levels == 4][layer] for tune 1 (OQ), delta_rate_sq[layer] for tune 0, delta_rate_vmaf[layer] for
 * tune 2 (EbRateControlProcess.c:34-41).  Key frames take the adaptive path (QP_SCALING_MODE_1: kf_boost from the picture
 * analysis' non-moving score, :4592-4650), which is rate control proper and NOT reproduced: is_key returns the sequence q index.
 */
#include "../../include/svtvp9_hip.h"

static double q_of(int qindex) { return svt_hip_vp9_ac_step(qindex) / 4.0; }

/* eb_vp9_compute_qdelta with best_quality = MINQ (0), worst_quality = MAXQ (255) */
static int compute_qdelta(double qstart, double qtarget) {
    int start_index = 255, target_index = 255;
    for (int i = 0; i < 255; ++i) { start_index = i; if (q_of(i) >= qstart) break; }
    for (int i = 0; i < 255; ++i) { target_index = i; if (q_of(i) >= qtarget) break; }
    return target_index - start_index;
}

int32_t svt_hip_vp9_layer_qindex(int32_t qp, int32_t tune, int32_t hierarchical_levels, int32_t temporal_layer_index, int32_t is_key) {
    static const double rate_oq[2][6] = {{0.35, 0.70, 0.85, 1.00, 1.00, 1.00}, {0.30, 0.6, 0.8, 0.9, 1.0, 1.0}};
    static const double rate_sq[6] = {0.35, 0.50, 0.75, 1.00, 1.00, 1.00}, rate_vmaf[6] = {0.50, 0.70, 0.85, 1.00, 1.00, 1.00};
    const int qindex = svt_hip_vp9_qindex_from_qp(qp);
    if (is_key || qindex < 0 || temporal_layer_index < 0 || temporal_layer_index > 5) return qindex;
    const double q = q_of(qindex);
    const double r = tune == 1 ? rate_oq[hierarchical_levels == 4][temporal_layer_index] : tune == 0 ? rate_sq[temporal_layer_index] : rate_vmaf[temporal_layer_index];
    const int    v = qindex + compute_qdelta(q, q * r);
    return v > 0 ? v : 0;
}
