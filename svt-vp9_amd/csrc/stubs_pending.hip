/* Entry points declared in include/svtvp9_hip.h whose kernels are not written yet: they fail loudly. */
#include "svt_ctx.h"
extern "C" void    svt_hip_lf_thresh_init(svt_lf_thresh *, int32_t) {}
extern "C" int32_t svt_hip_lf_level_from_q(int32_t, int32_t) { return -1; }
extern "C" int32_t svt_hip_lf_frame_device(svt_hip_ctx *, const svt_yuv_planes *, const svt_lf_mask *, int32_t, const svt_lf_thresh *,
                                           int32_t, int32_t, int32_t) {
    return svt_set_error(SVT_HIP_ERR_UNSUPPORTED, "lf: kernel not built");
}
extern "C" int32_t svt_hip_lf_frame(svt_hip_ctx *, const svt_yuv_planes *, const svt_lf_mask *, int32_t, const svt_lf_thresh *, int32_t,
                                    int32_t, int32_t) {
    return svt_set_error(SVT_HIP_ERR_UNSUPPORTED, "lf: kernel not built");
}
