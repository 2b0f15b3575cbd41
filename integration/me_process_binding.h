/*
 * integration/me_process_binding.h -- the reference-side binding of libsvtvp9_hip.so's motion estimation.
 *
 * What a maintainer of the reference adds to Source/Lib/Codec/EbMotionEstimationProcess.c: this header is written against the
 * reference's own structures (PictureParentControlSet, SequenceControlSet, MotionEstimationContext / MeContext, EbPaReferenceObject,
 * EbPictureBufferDesc, MeCuResults) and replaces the SB loop of eb_vp9_motion_estimation_kernel
 * (Codec/EbMotionEstimationProcess.c:964-1044: copy the SB into sb_buffer / the 1/4 and 1/16 SB buffers, motion_estimate_sb per SB)
 * by ONE call per picture.  It is compiled (-Wall -Werror) and EXECUTED against the reference's structures by oracle/ref_meproc_driver.c,
 * which runs the reference's own thread function on the same picture and compares picture_control_set_ptr->me_results byte for byte
 * (tests/test_ref_me_process.py).  INTEGRATION.md quotes this file.
 *
 * Include it after the reference's headers (EbPictureControlSet.h, EbSequenceControlSet.h, EbMotionEstimationProcess.h,
 * EbMotionEstimationContext.h, EbReferenceObject.h, EbPictureBufferDesc.h) and after svtvp9_hip.h.
 */
#ifndef SVT_HIP_ME_PROCESS_BINDING_H
#define SVT_HIP_ME_PROCESS_BINDING_H
#include <string.h>

/* per ME thread, beside its MotionEstimationContext (eb_vp9_motion_estimation_context_ctor, :326): the device context and the
 * host copies of a picture's results (n_sb * 85 records, n_sb distortions), allocated once */
typedef struct SvtHipMeBinding {
    svt_hip_ctx      *hip;
    svt_me_pu_result *results;
    uint32_t         *rcme;
    int32_t           n_sb;
} SvtHipMeBinding;

static inline void svt_hip_bind_plane(const EbPictureBufferDesc *d, svt_plane *p) { /* Codec/EbPictureBufferDesc.h:27-59 */
    p->buf      = d->buffer_y;
    p->stride   = d->stride_y;
    p->origin_x = d->origin_x;
    p->origin_y = d->origin_y;
    p->width    = d->width;
    p->height   = d->height;
}
static inline void svt_hip_bind_pa_picture(const EbPaReferenceObject *o, svt_pa_picture *pa) { /* Codec/EbReferenceObject.h:39-49 */
    svt_hip_bind_plane(o->input_padded_picture_ptr, &pa->full);
    svt_hip_bind_plane(o->quarter_decimated_picture_ptr, &pa->quarter);
    svt_hip_bind_plane(o->sixteenth_decimated_picture_ptr, &pa->sixteenth);
}

/* every field is a plain copy of what motion_estimate_sb reads from the control sets and the MeContext
 * (Codec/EbMotionEstimation.c:4584-4631); the MeContext fields are those the kernel's signal derivation has just set (:921-927) */
static inline void svt_hip_bind_me_params(const SequenceControlSet *sequence_control_set_ptr, const PictureParentControlSet *picture_control_set_ptr,
                                          const MeContext *mc, svt_me_params *p) {
    const int b_slice = picture_control_set_ptr->slice_type == B_SLICE;
    memset(p, 0, sizeof *p);
    p->num_ref_lists            = (uint8_t)(b_slice ? 2 : 1);
    p->temporal_layer_index     = picture_control_set_ptr->temporal_layer_index;
    p->hierarchical_levels      = picture_control_set_ptr->hierarchical_levels;
    p->enable_hme_flag          = picture_control_set_ptr->enable_hme_flag;
    p->enable_hme_level_0_flag  = picture_control_set_ptr->enable_hme_level_0_flag;
    p->enable_hme_level_1_flag  = picture_control_set_ptr->enable_hme_level_1_flag;
    p->enable_hme_level_2_flag  = picture_control_set_ptr->enable_hme_level_2_flag;
    p->cu8x8_mode               = (uint8_t)picture_control_set_ptr->cu8x8_mode;
    p->cu16x16_mode             = (uint8_t)picture_control_set_ptr->cu16x16_mode;
    p->same_ref_poc             = (uint8_t)(b_slice && picture_control_set_ptr->ref_pic_poc_array[0] == picture_control_set_ptr->ref_pic_poc_array[1]);
    p->rate_control_mode        = (uint8_t)sequence_control_set_ptr->static_config.rate_control_mode;
    p->fractional_search_method = (uint8_t)mc->fractional_search_method;
    p->fractional_search_model  = (uint8_t)mc->fractional_search_model;
    p->fractional_search64x64   = (uint8_t)mc->fractional_search64x64;
    p->single_hme_quadrant      = (uint8_t)mc->single_hme_quadrant;
    p->search_area_width        = mc->search_area_width;
    p->search_area_height       = mc->search_area_height;
    p->number_hme_search_region_in_width   = mc->number_hme_search_region_in_width;
    p->number_hme_search_region_in_height  = mc->number_hme_search_region_in_height;
    p->hme_level0_total_search_area_width  = mc->hme_level0_total_search_area_width;
    p->hme_level0_total_search_area_height = mc->hme_level0_total_search_area_height;
    for (int i = 0; i < 2; i++) {
        p->hme_level0_search_area_in_width_array[i]  = mc->hme_level0_search_area_in_width_array[i];
        p->hme_level0_search_area_in_height_array[i] = mc->hme_level0_search_area_in_height_array[i];
        p->hme_level1_search_area_in_width_array[i]  = mc->hme_level1_search_area_in_width_array[i];
        p->hme_level1_search_area_in_height_array[i] = mc->hme_level1_search_area_in_height_array[i];
        p->hme_level2_search_area_in_width_array[i]  = mc->hme_level2_search_area_in_width_array[i];
        p->hme_level2_search_area_in_height_array[i] = mc->hme_level2_search_area_in_height_array[i];
    }
}

/* Inside eb_vp9_motion_estimation_kernel, instead of the SB loop (:964-1044), behind the signal derivation (:921-927).  The reference
 * cuts a picture into segments taken by different threads; with the device path the task of segment 0 issues ONE call for the whole
 * picture and the other segment tasks skip the loop:
 *
 *     if (picture_control_set_ptr->slice_type != I_SLICE && segment_index == 0)
 *         svt_hip_bind_me_picture(binding, context_ptr, sequence_control_set_ptr, picture_control_set_ptr);
 *
 * Fills picture_control_set_ptr->me_results[sb][pu] for every SB and, under rate control, rcme_distortion[sb] (:5295-5302). */
static inline EbErrorType svt_hip_bind_me_picture(SvtHipMeBinding *b, MotionEstimationContext *context_ptr, SequenceControlSet *sequence_control_set_ptr,
                                                  PictureParentControlSet *picture_control_set_ptr) {
    svt_pa_picture cur, ref0, ref1;
    svt_me_params  p;
    const int      b_slice = picture_control_set_ptr->slice_type == B_SLICE;
    svt_hip_bind_pa_picture((const EbPaReferenceObject *)picture_control_set_ptr->pareference_picture_wrapper_ptr->object_ptr, &cur);
    svt_hip_bind_pa_picture((const EbPaReferenceObject *)picture_control_set_ptr->ref_pa_pic_ptr_array[0]->object_ptr, &ref0);
    if (b_slice) svt_hip_bind_pa_picture((const EbPaReferenceObject *)picture_control_set_ptr->ref_pa_pic_ptr_array[1]->object_ptr, &ref1);
    /* the picture is the sequence's luma size (the planes of an EbPaReferenceObject are allocated for the maximum size) */
    cur.full.width = ref0.full.width = sequence_control_set_ptr->luma_width;
    cur.full.height = ref0.full.height = sequence_control_set_ptr->luma_height;
    if (b_slice) { ref1.full.width = cur.full.width; ref1.full.height = cur.full.height; }
    svt_hip_bind_me_params(sequence_control_set_ptr, picture_control_set_ptr, context_ptr->me_context_ptr, &p);

    const int32_t n_sb = svt_hip_sb_count(cur.full.width, cur.full.height);
    if (n_sb > b->n_sb) return EB_ErrorBadParameter;
    if (svt_hip_me_picture(b->hip, &cur, &ref0, b_slice ? &ref1 : NULL, &p, b->results, b->rcme) != 0)
        return EB_ErrorMax; /* svt_hip_last_error() has the text: a bad argument or a device failure (every search method is offloaded) */
    /* svt_me_pu_result has MeCuResults' field order (Codec/EbMotionEstimationLcuResults.h:41-54); `direction` is a 2-bit field there */
    for (int32_t sb = 0; sb < n_sb; sb++)
        for (int pu = 0; pu < SQUARE_PU_COUNT; pu++) {
            const svt_me_pu_result *r = &b->results[sb * 85 + pu];
            MeCuResults            *m = &picture_control_set_ptr->me_results[sb][pu];
            m->x_mv_l0 = r->x_mv_l0; m->y_mv_l0 = r->y_mv_l0; m->x_mv_l1 = r->x_mv_l1; m->y_mv_l1 = r->y_mv_l1;
            for (int k = 0; k < 3; k++) {
                m->distortion_direction[k].distortion = r->distortion_direction[k].distortion;
                m->distortion_direction[k].direction  = r->distortion_direction[k].direction;
            }
            m->total_me_candidate_index = r->total_me_candidate_index;
        }
    if (p.rate_control_mode)
        for (int32_t sb = 0; sb < n_sb; sb++) picture_control_set_ptr->rcme_distortion[sb] = b->rcme[sb];
    return EB_ErrorNone;
}
#endif
