/*
 * gop_shard.c -- host-side (plain C) work distribution of the hot path over the GPUs of a node (scope row e).
 * Closed GOPs share nothing (Codec/EbPictureDecisionProcess.c:952, 1596-1603): round-robin by GOP index; a split GOP hands one
 * padded reference picture from a mini-GOP's device to the next one's (svt_hip_ref_handoff_device).
 */
#include "../../include/svtvp9_hip.h"

int32_t svt_hip_gop_owner(int64_t gop, int32_t n_devices) { return n_devices > 0 && gop >= 0 ? (int32_t)(gop % n_devices) : -1; }

int32_t svt_hip_gop_assign(int64_t n_gops, int32_t n_devices, int32_t index, int64_t *gops, int32_t max) {
    if (n_gops < 0 || n_devices < 1 || index < 0 || index >= n_devices || (max > 0 && !gops)) return SVT_HIP_ERR_BAD_PARAMETER;
    int32_t n = 0;
    for (int64_t g = index; g < n_gops; g += n_devices) {
        if (n < max) gops[n] = g;
        n++;
    }
    return n;
}

int32_t svt_hip_minigop_reference_source(int64_t minigop, int32_t n_devices) {
    if (n_devices < 1 || minigop <= 0) return -1;
    return (int32_t)((minigop - 1) % n_devices);
}
