/*
 * integration/loop_filter_binding.h -- the reference-side binding of libsvtvp9_hip.so's deblocking.
 *
 * What a maintainer of the reference adds at the call site Source/Lib/Codec/EbEncDecProcess.c:5676-5686 (the picture's last SB in
 * eb_vp9_enc_dec_kernel): eb_vp9_build_mask_frame stays (it fills cm->lf.lfm[] and cm->lf_info), the call of eb_vp9_loop_filter_frame
 * (VPX/vp9_loopfilter.c:1521) is replaced by svt_hip_bind_loop_filter_frame with the SAME arguments.  Written against the reference's
 * VP9_COMMON / MACROBLOCKD / LOOP_FILTER_MASK / loop_filter_info_n; compiled (-Wall -Werror) and EXECUTED against them by
 * oracle/ref_lfbind_driver.c, which runs the reference's own two calls on the same frame and compares every sample
 * (tests/test_ref_lf_call_site.py).  INTEGRATION.md quotes this file.
 *
 * Include it after the reference's headers (vp9_onyxc_int.h, vp9_blockd.h, vp9_loopfilter.h) and after svtvp9_hip.h.
 */
#ifndef SVT_HIP_LOOP_FILTER_BINDING_H
#define SVT_HIP_LOOP_FILTER_BINDING_H

/* LOOP_FILTER_MASK (VPX/vp9_loopfilter.h:82-95) and svt_lf_mask are the same 160 bytes: the mask array is passed as it is */
typedef char svt_hip_lf_mask_layout_check[sizeof(LOOP_FILTER_MASK) == sizeof(svt_lf_mask) ? 1 : -1];

/* same arguments and the same early exits as eb_vp9_loop_filter_frame; `partial_frame` is never set by the reference's encode pass
 * (:5686 passes 0) and is refused here */
static inline int svt_hip_bind_loop_filter_frame(svt_hip_ctx *hip, VP9_COMMON *cm, MACROBLOCKD *xd, int frame_filter_level, int y_only, int partial_frame) {
    if (!frame_filter_level) return 0;
    if (partial_frame) return -4; /* SVT_HIP_ERR_UNSUPPORTED */
    /* the threshold tables eb_vp9_loop_filter_frame_init (called by eb_vp9_build_mask_frame) left in cm->lf_info for the picture's
       sharpness: lfthr[level].{mblim, lim, hev_thr} are SIMD_WIDTH copies of one byte each (VPX/vp9_loopfilter.h:64-68) */
    svt_lf_thresh thr;
    for (int lvl = 0; lvl <= MAX_LOOP_FILTER; lvl++) {
        thr.mblim[lvl]   = cm->lf_info.lfthr[lvl].mblim[0];
        thr.lim[lvl]     = cm->lf_info.lfthr[lvl].lim[0];
        thr.hev_thr[lvl] = cm->lf_info.lfthr[lvl].hev_thr[0];
    }
    svt_yuv_planes rec;
    rec.y = xd->plane[0].dst.buf; rec.u = xd->plane[1].dst.buf; rec.v = xd->plane[2].dst.buf;
    rec.y_stride = xd->plane[0].dst.stride; rec.uv_stride = xd->plane[1].dst.stride;
    rec.width = cm->mi_cols * MI_SIZE; rec.height = cm->mi_rows * MI_SIZE;
    return svt_hip_lf_frame(hip, &rec, (const svt_lf_mask *)cm->lf.lfm, cm->lf.lfm_stride, &thr, cm->mi_rows, cm->mi_cols, y_only);
}
#endif
