/*
 * me_side.c -- host-side (plain C) per-SB side decisions of the ME stage that are scalar formulas on picture-analysis
 * statistics: eb_vp9_derive_similar_collocated_flag (Codec/EbMotionEstimationProcess.c:747-783).
 */
#include "../../include/svtvp9_hip.h"

void svt_hip_me_similar_collocated(const uint8_t *cur_mean, const uint16_t *cur_var, const uint8_t *ref_mean,
                                   const uint16_t *ref_var, int32_t n_sb, int32_t is_i_slice,
                                   int32_t is_used_as_reference, uint8_t *similar, uint8_t *similar_all_layers) {
    for (int32_t sb = 0; sb < n_sb; sb++) {
        uint8_t s = 0, a = 0;
        if (!is_i_slice) {
            int64_t rv = ref_var[sb] ? ref_var[sb] : 1; /* MAX(ref_var, 1) */
            int64_t dm = (int64_t)cur_mean[sb] - (int64_t)ref_mean[sb];
            int64_t pr = (int64_t)cur_var[sb] * 100 / rv - 100, dv = (int64_t)cur_var[sb] - rv;
            if (dm < 0) dm = -dm;
            if (pr < 0) pr = -pr;
            if (dv < 0) dv = -dv;
            if (dm < 10 && (pr < 10 || dv < 10)) { /* MEAN_DIFF_THRSHOLD, VAR_DIFF_THRSHOLD (Codec/EbDefinitions.h:687-688) */
                s = is_used_as_reference ? 1 : 0;
                a = 1;
            }
        }
        similar[sb] = s;
        similar_all_layers[sb] = a;
    }
}
