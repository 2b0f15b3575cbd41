/*
 * ref_mepreset_driver.c -- harness that runs the REFERENCE's ME parameter derivation for a list of (picture size, enc_mode,
 * tune, temporal layer, ...) tuples and prints what motion_estimate_sb would then read.  TEST INFRASTRUCTURE ONLY (rules:
 * ref_me_driver.c).  Functions called, all from the reference's own objects:
 *   eb_vp9_derive_input_resolution                       (Codec/EbSequenceControlSet.c:489)
 *   eb_vp9_signal_derivation_pre_analysis_{sq,oq,vmaf}   (Codec/EbResourceCoordinationProcess.c:291-460; HME enables)
 *   eb_vp9_signal_derivation_multi_processes_{sq,oq,vmaf}(Codec/EbPictureDecisionProcess.c:755-925; use_subpel_flag, cu8x8_mode)
 *   eb_vp9_signal_derivation_me_kernel_{sq,oq,vmaf}      (Codec/EbMotionEstimationProcess.c:541-720; search areas, method)
 *
 * stdin : lines "W H enc_mode tune temporal_layer is_used_as_reference frame_rate"
 * stdout: per line 29 integers: input_resolution enable_hme enable_l0 enable_l1 enable_l2 use_subpel cu8x8_mode cu16x16_mode
 *         single_hme_quadrant fractional_search_method fractional_search64x64 fractional_search_model search_area_width
 *         search_area_height n_region_w n_region_h l0_total_w l0_total_h l0_w[2] l0_h[2] l1_w[2] l1_h[2] l2_w[2] l2_h[2]
 *         (29 numbers)
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>

#include "EbDefinitions.h"
#include "EbPictureControlSet.h"
#include "EbSequenceControlSet.h"
#include "EbMotionEstimationProcess.h"
#include "EbMotionEstimationContext.h"

uint32_t eb_vp9_ASM_TYPES = 0;

EbErrorType eb_vp9_signal_derivation_pre_analysis_sq(SequenceControlSet *, PictureParentControlSet *);
EbErrorType eb_vp9_signal_derivation_pre_analysis_oq(SequenceControlSet *, PictureParentControlSet *);
EbErrorType eb_vp9_signal_derivation_pre_analysis_vmaf(SequenceControlSet *, PictureParentControlSet *);
EbErrorType eb_vp9_signal_derivation_multi_processes_sq(SequenceControlSet *, PictureParentControlSet *);
EbErrorType eb_vp9_signal_derivation_multi_processes_oq(SequenceControlSet *, PictureParentControlSet *);
EbErrorType eb_vp9_signal_derivation_multi_processes_vmaf(SequenceControlSet *, PictureParentControlSet *);
EbErrorType eb_vp9_signal_derivation_me_kernel_sq(SequenceControlSet *, PictureParentControlSet *, MotionEstimationContext *);
EbErrorType eb_vp9_signal_derivation_me_kernel_oq(SequenceControlSet *, PictureParentControlSet *, MotionEstimationContext *);
EbErrorType eb_vp9_signal_derivation_me_kernel_vmaf(SequenceControlSet *, PictureParentControlSet *, MotionEstimationContext *);

int main(void) {
    SequenceControlSet      *scs = (SequenceControlSet *)calloc(1, sizeof *scs);
    PictureParentControlSet *pcs = (PictureParentControlSet *)calloc(1, sizeof *pcs);
    MotionEstimationContext *mec = (MotionEstimationContext *)calloc(1, sizeof *mec);
    MeContext               *me  = (MeContext *)calloc(1, sizeof *me);
    mec->me_context_ptr = me;
    int W, H, mode, tune, tl, used, fps;
    while (scanf("%d %d %d %d %d %d %d", &W, &H, &mode, &tune, &tl, &used, &fps) == 7) {
        memset(me, 0xEE, sizeof *me);
        scs->luma_width = (uint16_t)W; scs->luma_height = (uint16_t)H;
        scs->static_config.use_default_me_hme = 1;
        scs->static_config.frame_rate = (uint32_t)fps << 16;
        scs->static_config.tune = (uint8_t)tune;
        eb_vp9_derive_input_resolution(scs, (uint32_t)W * (uint32_t)H);
        pcs->enc_mode = (uint8_t)mode;
        pcs->temporal_layer_index = (uint8_t)tl;
        pcs->is_used_as_reference_flag = (EB_BOOL)used;
        pcs->slice_type = B_SLICE;
        if (tune == TUNE_SQ) {
            eb_vp9_signal_derivation_pre_analysis_sq(scs, pcs);
            eb_vp9_signal_derivation_multi_processes_sq(scs, pcs);
            eb_vp9_signal_derivation_me_kernel_sq(scs, pcs, mec);
        } else if (tune == TUNE_VMAF) {
            eb_vp9_signal_derivation_pre_analysis_vmaf(scs, pcs);
            eb_vp9_signal_derivation_multi_processes_vmaf(scs, pcs);
            eb_vp9_signal_derivation_me_kernel_vmaf(scs, pcs, mec);
        } else {
            eb_vp9_signal_derivation_pre_analysis_oq(scs, pcs);
            eb_vp9_signal_derivation_multi_processes_oq(scs, pcs);
            eb_vp9_signal_derivation_me_kernel_oq(scs, pcs, mec);
        }
        printf("%d %d %d %d %d %d %d %d %d %d %d %d %d %d %d %d %d %d", scs->input_resolution, pcs->enable_hme_flag,
               pcs->enable_hme_level_0_flag, pcs->enable_hme_level_1_flag, pcs->enable_hme_level_2_flag, pcs->use_subpel_flag,
               pcs->cu8x8_mode, pcs->cu16x16_mode, me->single_hme_quadrant, me->fractional_search_method,
               me->fractional_search64x64, me->fractional_search_model, me->search_area_width, me->search_area_height,
               me->number_hme_search_region_in_width, me->number_hme_search_region_in_height,
               me->hme_level0_total_search_area_width, me->hme_level0_total_search_area_height);
        for (int i = 0; i < 2; i++) printf(" %d", me->hme_level0_search_area_in_width_array[i]);
        for (int i = 0; i < 2; i++) printf(" %d", me->hme_level0_search_area_in_height_array[i]);
        for (int i = 0; i < 2; i++) printf(" %d", me->hme_level1_search_area_in_width_array[i]);
        for (int i = 0; i < 2; i++) printf(" %d", me->hme_level1_search_area_in_height_array[i]);
        for (int i = 0; i < 2; i++) printf(" %d", me->hme_level2_search_area_in_width_array[i]);
        for (int i = 0; i < 2; i++) printf(" %d", me->hme_level2_search_area_in_height_array[i]);
        printf("\n");
    }
    return 0;
}
