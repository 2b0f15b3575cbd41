/*
 * ref_meproc_driver.c -- harness that runs the REFERENCE's motion-estimation kernel process ITSELF,
 * eb_vp9_motion_estimation_kernel (Source/Lib/Codec/EbMotionEstimationProcess.c:875-1290: signal derivation, the SB loop :964-1044,
 * similar-collocated / stationary-edge decisions, the rate-control SAD-interval histograms :1103-1237), as a thread fed through the
 * reference's own system-resource FIFOs (Codec/EbSystemResourceManager.c, Codec/EbThreads.c), and then the BINDING of this repository
 * (integration/me_process_binding.h -> svt_hip_me_picture) on the same PictureParentControlSet / EbPaReferenceObjects / MeContext.
 * TEST INFRASTRUCTURE ONLY (rules: ref_me_driver.c).  No reference code is copied: the reference file is compiled as part of this
 * translation unit where it lies (as ref_meside_driver.c does), the control-set objects are allocated here and filled with what the
 * upstream processes would have left in them (the picture-level flags by the reference's own derivation functions).
 *
 * request : int32 magic 'SVMP', W, H, enc_mode, tune, temporal_layer, slice (0 B, 1 P), is_used_as_reference, rate_control_mode,
 *           same_ref_poc, seg_cols, seg_rows, run_binding, device;  then 9 planes (cur.full, cur.quarter, cur.sixteenth, ref0.*, ref1.*),
 *           each int32 stride, origin_x, origin_y, width, height, nbytes + bytes;  then per SB: uint8 cur_mean64, uint16 var[5]
 *           (64x64, the four 32x32), uint8 ref_mean, uint16 ref_var
 * response: int32 n_sb, NUMBER_OF_SAD_INTERVALS, NUMBER_OF_INTRA_SAD_INTERVALS;  REFERENCE: svt_me_pu_result[n_sb * 85], uint32 rcme[n_sb], uint32 inter_sad_interval_index[n_sb],
 *           uint32 intra_sad_interval_index[n_sb], uint16 me_distortion_histogram[NUMBER_OF_SAD_INTERVALS],
 *           uint16 ois_distortion_histogram[NUMBER_OF_INTRA_SAD_INTERVALS], uint32 full_sb_count, uint8 similar[n_sb],
 *           similar_all_layers[n_sb], check1[n_sb], pm_check1[n_sb];
 *           svt_me_params as the binding derived it;  int32 binding_rc;  BINDING (when run_binding): svt_me_pu_result[n_sb * 85]
 *           (read back from me_results), uint32 rcme[n_sb]
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <unistd.h>

#include "EbDefinitions.h"
#include "EbPictureControlSet.h"
#include "EbSequenceControlSet.h"
#include "EbMotionEstimationProcess.h"
#include "EbMotionEstimationContext.h"
#include "EbMotionEstimationResults.h"
#include "EbPictureDecisionResults.h"
#include "EbReferenceObject.h"
#include "EbPictureBufferDesc.h"
#include "EbSystemResourceManager.h"
#include "EbThreads.h"

/* the globals Codec/EbEncHandle.c defines in the encoder: the dispatch selector and the allocation-tracking table of EB_MALLOC */
uint32_t          eb_vp9_ASM_TYPES = 0;
EbMemoryMapEntry *memory_map       = 0;
uint32_t         *memory_map_index = 0;
uint64_t         *total_lib_memory = 0;
uint32_t          lib_malloc_count = 0;
uint32_t          lib_thread_count = 0, lib_mutex_count = 0, lib_semaphore_count = 0; /* (counters of EB_CREATETHREAD / _MUTEX / _SEMAPHORE) */
static uint32_t   map_index_storage;
static uint64_t   total_memory_storage;

/* the reference's file itself (thread function, signal derivation, static helpers) */
#include "EbMotionEstimationProcess.c"

EbErrorType eb_vp9_signal_derivation_pre_analysis_sq(SequenceControlSet *, PictureParentControlSet *);
EbErrorType eb_vp9_signal_derivation_pre_analysis_oq(SequenceControlSet *, PictureParentControlSet *);
EbErrorType eb_vp9_signal_derivation_pre_analysis_vmaf(SequenceControlSet *, PictureParentControlSet *);
EbErrorType eb_vp9_signal_derivation_multi_processes_sq(SequenceControlSet *, PictureParentControlSet *);
EbErrorType eb_vp9_signal_derivation_multi_processes_oq(SequenceControlSet *, PictureParentControlSet *);
EbErrorType eb_vp9_signal_derivation_multi_processes_vmaf(SequenceControlSet *, PictureParentControlSet *);

/* ---- this repository's side: the C ABI and the binding a maintainer adds to the reference ---- */
#include "../include/svtvp9_hip.h"
#include "../integration/me_process_binding.h"

static int rd(FILE *f, void *p, size_t n) { return fread(p, 1, n, f) == n ? 0 : -1; }
static EbPictureBufferDesc *read_plane(FILE *f) {
    int32_t              h[6];
    EbPictureBufferDesc *d = (EbPictureBufferDesc *)calloc(1, sizeof *d);
    if (rd(f, h, sizeof h)) exit(3);
    d->stride_y = (uint16_t)h[0]; d->origin_x = (uint16_t)h[1]; d->origin_y = (uint16_t)h[2];
    d->width = (uint16_t)h[3]; d->height = (uint16_t)h[4];
    if (h[5] > 0) {
        d->buffer_y = (uint8_t *)malloc((size_t)h[5] + 4096);
        if (rd(f, d->buffer_y, (size_t)h[5])) exit(3);
    }
    return d;
}
static void dump_me_results(FILE *o, PictureParentControlSet *pcs, int n_sb) {
    for (int sb = 0; sb < n_sb; sb++)
        for (int pu = 0; pu < SQUARE_PU_COUNT; pu++) {
            const MeCuResults *m = &pcs->me_results[sb][pu];
            svt_me_pu_result   r;
            memset(&r, 0, sizeof r);
            r.x_mv_l0 = m->x_mv_l0; r.y_mv_l0 = m->y_mv_l0; r.x_mv_l1 = m->x_mv_l1; r.y_mv_l1 = m->y_mv_l1;
            for (int k = 0; k < 3; k++) { r.distortion_direction[k].distortion = m->distortion_direction[k].distortion; r.distortion_direction[k].direction = m->distortion_direction[k].direction; }
            r.total_me_candidate_index = m->total_me_candidate_index;
            fwrite(&r, sizeof r, 1, o);
        }
}

int main(int argc, char **argv) {
    if (argc < 3) { fprintf(stderr, "usage: %s request.bin response.bin\n", argv[0]); return 2; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    int32_t h[15];
    if (rd(f, h, sizeof h) || h[0] != 0x504D5653) return 3; /* 'SVMP' */
    const int W = h[1], H = h[2], enc_mode = h[3], tune = h[4], tl = h[5], p_slice = h[6], used = h[7], rc_mode = h[8], same_poc = h[9];
    const int seg_cols = h[10] > 0 ? h[10] : 1, seg_rows = h[11] > 0 ? h[11] : 1, run_binding = h[12], device = h[13];
    EbPictureBufferDesc *pl[9];
    for (int i = 0; i < 9; i++) pl[i] = read_plane(f);
    const int nx = (W + 63) / 64, ny = (H + 63) / 64, n_sb = nx * ny;

    memory_map = (EbMemoryMapEntry *)calloc(1 << 16, sizeof *memory_map);
    memory_map_index = &map_index_storage; total_lib_memory = &total_memory_storage;

    /* ---- sequence control set: what the resource-coordination process leaves in it ---- */
    SequenceControlSet *scs = (SequenceControlSet *)calloc(1, sizeof *scs);
    scs->luma_width = (uint16_t)W; scs->luma_height = (uint16_t)H;
    scs->static_config.use_default_me_hme = 1;
    scs->static_config.frame_rate         = 60u << 16;
    scs->static_config.tune               = (uint8_t)tune;
    scs->static_config.rate_control_mode  = (uint32_t)rc_mode;
    scs->static_config.enc_mode           = (uint8_t)enc_mode;
    scs->look_ahead_distance              = 0; /* the ZZ-SAD / stationary-edge part 2 paths are pinned by ref_meside_driver.c */
    eb_vp9_derive_input_resolution(scs, (uint32_t)W * (uint32_t)H);
    if (eb_vp9_sb_params_init(scs) != EB_ErrorNone) return 4;
    EbObjectWrapper scs_w; memset(&scs_w, 0, sizeof scs_w); scs_w.object_ptr = scs;

    /* ---- analysed pictures (EbPaReferenceObject: the three planes + the per-SB statistics the collocated check reads) ---- */
    EbPaReferenceObject *cur_obj = (EbPaReferenceObject *)calloc(1, sizeof *cur_obj), *ref_obj = (EbPaReferenceObject *)calloc(2, sizeof *ref_obj);
    cur_obj->input_padded_picture_ptr = pl[0]; cur_obj->quarter_decimated_picture_ptr = pl[1]; cur_obj->sixteenth_decimated_picture_ptr = pl[2];
    for (int l = 0; l < 2; l++) { /* (y_mean[] / variance[] are arrays inside the object: filled per SB below) */
        ref_obj[l].input_padded_picture_ptr = pl[3 + 3 * l]; ref_obj[l].quarter_decimated_picture_ptr = pl[4 + 3 * l]; ref_obj[l].sixteenth_decimated_picture_ptr = pl[5 + 3 * l];
    }
    EbObjectWrapper cur_w, ref_w[2];
    memset(&cur_w, 0, sizeof cur_w); memset(ref_w, 0, sizeof ref_w);
    cur_w.object_ptr = cur_obj; ref_w[0].object_ptr = &ref_obj[0]; ref_w[1].object_ptr = &ref_obj[1];

    /* ---- picture control set ---- */
    PictureParentControlSet *pcs = (PictureParentControlSet *)calloc(1, sizeof *pcs);
    pcs->sequence_control_set_wrapper_ptr = &scs_w;
    pcs->pareference_picture_wrapper_ptr  = &cur_w;
    pcs->enhanced_picture_ptr             = pl[0]; /* the input picture: same luma samples as its padded analysis copy */
    pcs->ref_pa_pic_ptr_array[0] = &ref_w[0]; pcs->ref_pa_pic_ptr_array[1] = &ref_w[p_slice ? 0 : 1];
    pcs->ref_pic_poc_array[0] = 8; pcs->ref_pic_poc_array[1] = same_poc ? 8 : 16;
    pcs->slice_type = p_slice ? P_SLICE : B_SLICE;
    pcs->temporal_layer_index = (uint8_t)tl; pcs->hierarchical_levels = (uint8_t)(tune == TUNE_SQ ? 3 : 4);
    pcs->enc_mode = (uint8_t)enc_mode; pcs->is_used_as_reference_flag = (EB_BOOL)used;
    pcs->picture_number = 8; pcs->end_of_sequence_flag = EB_FALSE;
    pcs->max_number_of_pus_per_sb = SQUARE_PU_COUNT;
    pcs->me_segments_column_count = (uint8_t)seg_cols; pcs->me_segments_row_count = (uint8_t)seg_rows;
    pcs->me_segments_total_count = (uint16_t)(seg_cols * seg_rows);
    pcs->me_results = (MeCuResults **)calloc((size_t)n_sb, sizeof(MeCuResults *));
    pcs->variance = (uint16_t **)calloc((size_t)n_sb, sizeof(uint16_t *));
    pcs->y_mean = (uint8_t **)calloc((size_t)n_sb, sizeof(uint8_t *));
    for (int i = 0; i < n_sb; i++) {
        pcs->me_results[i] = (MeCuResults *)calloc(SQUARE_PU_COUNT, sizeof(MeCuResults));
        pcs->variance[i] = (uint16_t *)calloc(MAX_ME_PU_COUNT, 2); pcs->y_mean[i] = (uint8_t *)calloc(MAX_ME_PU_COUNT, 1);
        uint8_t  m8[2];
        uint16_t v[5], rv;
        if (rd(f, &m8[0], 1) || rd(f, v, sizeof v) || rd(f, &m8[1], 1) || rd(f, &rv, 2)) return 3;
        pcs->y_mean[i][PA_RASTER_SCAN_CU_INDEX_64x64] = m8[0];
        pcs->variance[i][PA_RASTER_SCAN_CU_INDEX_64x64] = v[0];
        for (int k = 0; k < 4; k++) pcs->variance[i][ME_TIER_ZERO_PU_32x32_0 + k] = v[1 + k];
        ref_obj[0].y_mean[i] = m8[1]; ref_obj[0].variance[i] = rv;
    }
    fclose(f);
    pcs->rcme_distortion = (uint32_t *)calloc((size_t)n_sb, sizeof(uint32_t));
    pcs->inter_sad_interval_index = (uint32_t *)calloc((size_t)n_sb, sizeof(uint32_t));
    pcs->intra_sad_interval_index = (uint32_t *)calloc((size_t)n_sb, sizeof(uint32_t));
    pcs->me_distortion_histogram = (uint16_t *)calloc(NUMBER_OF_SAD_INTERVALS, sizeof(uint16_t));
    pcs->ois_distortion_histogram = (uint16_t *)calloc(NUMBER_OF_INTRA_SAD_INTERVALS, sizeof(uint16_t));
    pcs->similar_colocated_sb_array = (EB_BOOL *)calloc((size_t)n_sb, sizeof(EB_BOOL));
    pcs->similar_colocated_sb_array_all_layers = (EB_BOOL *)calloc((size_t)n_sb, sizeof(EB_BOOL));
    pcs->sb_stat_array = (SbStat *)calloc((size_t)n_sb, sizeof(SbStat));
    pcs->rc_distortion_histogram_mutex = eb_vp9_create_mutex();
    /* the picture-level flags by the reference's own derivations (resource coordination, picture decision) */
    if (tune == TUNE_SQ) { eb_vp9_signal_derivation_pre_analysis_sq(scs, pcs); eb_vp9_signal_derivation_multi_processes_sq(scs, pcs); }
    else if (tune == TUNE_VMAF) { eb_vp9_signal_derivation_pre_analysis_vmaf(scs, pcs); eb_vp9_signal_derivation_multi_processes_vmaf(scs, pcs); }
    else { eb_vp9_signal_derivation_pre_analysis_oq(scs, pcs); eb_vp9_signal_derivation_multi_processes_oq(scs, pcs); }
    EbObjectWrapper pcs_w; memset(&pcs_w, 0, sizeof pcs_w); pcs_w.object_ptr = pcs;

    /* ---- the two FIFOs of the process and its context, built by the reference's constructors (Codec/EbEncHandle.c:1171-1205, 1495-1504) ---- */
    EbSystemResource *in_res = NULL, *out_res = NULL;
    EbFifo          **in_prod = NULL, **in_cons = NULL, **out_prod = NULL, **out_cons = NULL;
    PictureDecisionResultInitData   in_init;
    MotionEstimationResultsInitData out_init;
    if (eb_vp9_system_resource_ctor(&in_res, 16, 1, 1, &in_prod, &in_cons, EB_TRUE, eb_vp9_picture_decision_result_ctor, &in_init) != EB_ErrorNone) return 4;
    if (eb_vp9_system_resource_ctor(&out_res, 16, 1, 1, &out_prod, &out_cons, EB_TRUE, eb_vp9_motion_estimation_results_ctor, &out_init) != EB_ErrorNone) return 4;
    MotionEstimationContext *mec = NULL;
    if (eb_vp9_motion_estimation_context_ctor(&mec, in_cons[0], out_prod[0]) != EB_ErrorNone) return 4;
    if (!eb_vp9_create_thread(eb_vp9_motion_estimation_kernel, mec)) return 4;

    /* one task per segment, as the picture-decision process posts them (Codec/EbPictureDecisionProcess.c:1896-1912) */
    const int n_seg = seg_cols * seg_rows;
    for (int s = 0; s < n_seg; s++) {
        EbObjectWrapper *w = NULL;
        eb_vp9_get_empty_object(in_prod[0], &w);
        PictureDecisionResults *r = (PictureDecisionResults *)w->object_ptr;
        r->picture_control_set_wrapper_ptr = &pcs_w; r->segment_index = (uint32_t)s;
        eb_vp9_post_full_object(w);
    }
    for (int s = 0; s < n_seg; s++) { /* ... and the consumer (initial rate control) takes one result per segment */
        EbObjectWrapper *w = NULL;
        eb_vp9_get_full_object(out_cons[0], &w);
        eb_vp9_release_object(w);
    }

    FILE *o = fopen(argv[2], "wb");
    if (!o) return 2;
    const int32_t n32 = n_sb, nsad = NUMBER_OF_SAD_INTERVALS, nint = NUMBER_OF_INTRA_SAD_INTERVALS;
    fwrite(&n32, 4, 1, o); fwrite(&nsad, 4, 1, o); fwrite(&nint, 4, 1, o);
    dump_me_results(o, pcs, n_sb);
    fwrite(pcs->rcme_distortion, 4, (size_t)n_sb, o);
    fwrite(pcs->inter_sad_interval_index, 4, (size_t)n_sb, o);
    fwrite(pcs->intra_sad_interval_index, 4, (size_t)n_sb, o);
    fwrite(pcs->me_distortion_histogram, 2, NUMBER_OF_SAD_INTERVALS, o);
    fwrite(pcs->ois_distortion_histogram, 2, NUMBER_OF_INTRA_SAD_INTERVALS, o);
    { const uint32_t full = pcs->full_sb_count; fwrite(&full, 4, 1, o); }
    for (int i = 0; i < n_sb; i++) { const uint8_t b = (uint8_t)pcs->similar_colocated_sb_array[i]; fwrite(&b, 1, 1, o); }
    for (int i = 0; i < n_sb; i++) { const uint8_t b = (uint8_t)pcs->similar_colocated_sb_array_all_layers[i]; fwrite(&b, 1, 1, o); }
    for (int i = 0; i < n_sb; i++) { const uint8_t b = pcs->sb_stat_array[i].check1_for_logo_stationary_edge_over_time_flag; fwrite(&b, 1, 1, o); }
    for (int i = 0; i < n_sb; i++) { const uint8_t b = pcs->sb_stat_array[i].pm_check1_for_logo_stationary_edge_over_time_flag; fwrite(&b, 1, 1, o); }

    /* ---- the binding, where the reference calls: same control sets, the MeContext as the kernel's signal derivation left it ---- */
    svt_me_params bp;
    svt_hip_bind_me_params(scs, pcs, mec->me_context_ptr, &bp);
    fwrite(&bp, sizeof bp, 1, o);
    int32_t brc = -100;
    if (run_binding) {
        SvtHipMeBinding b;
        memset(&b, 0, sizeof b);
        for (int i = 0; i < n_sb; i++) memset(pcs->me_results[i], 0xA5, SQUARE_PU_COUNT * sizeof(MeCuResults));
        memset(pcs->rcme_distortion, 0xA5, (size_t)n_sb * 4);
        b.n_sb = n_sb; b.results = (svt_me_pu_result *)calloc((size_t)n_sb * 85, sizeof(svt_me_pu_result)); b.rcme = (uint32_t *)calloc((size_t)n_sb, 4);
        if (svt_hip_ctx_create(&b.hip, device) != 0) { fprintf(stderr, "binding: %s\n", svt_hip_last_error()); brc = -101; }
        else {
            brc = (int32_t)svt_hip_bind_me_picture(&b, mec, scs, pcs);
            if (brc) fprintf(stderr, "binding: %s\n", svt_hip_last_error());
            svt_hip_ctx_destroy(b.hip);
        }
        fwrite(&brc, 4, 1, o);
        dump_me_results(o, pcs, n_sb);
        fwrite(pcs->rcme_distortion, 4, (size_t)n_sb, o);
    } else fwrite(&brc, 4, 1, o);
    fclose(o);
    fflush(NULL);
    _exit(0); /* the kernel thread blocks on its input FIFO for ever, as in the encoder */
}
