/*
 * svtvp9_hip.h -- C-ABI of the MI355X (gfx950) implementation of SVT-VP9's block-level DSP hot path.
 *
 * Plain C, plain pointers and sizes; no C++/torch types.  This is the boundary a maintainer of the
 * reference binds: three *batched* entry points that replace the places where the
 * reference calls its per-block x86 kernels, plus context management.
 *
 *   svt_hip_me_picture   replaces the SB loop of eb_vp9_motion_estimation_kernel
 *                        (Source/Lib/Codec/EbMotionEstimationProcess.c:964-1044 -> motion_estimate_sb,
 *                         Source/Lib/Codec/EbMotionEstimation.c:4524-5305)
 *   svt_hip_tq_batch     replaces perform_coding_loop's residual -> fwd txfm -> quant -> (inv txfm + recon)
 *                        (Source/Lib/Codec/EbEncDecProcess.c:365-587; kernels VPX/fwd_txfm.c, vp9_dct.c,
 *                         quantize.c, inv_txfm.c, vp9_idct.c)
 *   svt_hip_lf_frame     replaces eb_vp9_loop_filter_frame (Source/Lib/VPX/vp9_loopfilter.c:1521,
 *                        called at Source/Lib/Codec/EbEncDecProcess.c:5678)
 *
 * All functions return 0 on success or a negative svt_hip_status.  Host buffers are caller-owned;
 * device buffers are owned by the context.  A context is bound to one HIP device and one stream;
 * calls on different contexts are independent (the reference runs up to 20 ME threads).
 *
 * Every pointer parameter named d_* is a DEVICE pointer (HBM resident); the *_host convenience
 * wrappers take host pointers and stage through the context's own device buffers.
 */
#ifndef SVTVP9_HIP_H
#define SVTVP9_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------------ */
/* status codes (negative = error; values mirror the sign convention of EbErrorType,                 */
/* Source/API/EbSvtVp9Enc.h:100-120, without reusing its 0x8000xxxx numbering)                        */
/* ------------------------------------------------------------------------------------------------ */
typedef enum svt_hip_status {
    SVT_HIP_OK                 = 0,
    SVT_HIP_ERR_BAD_PARAMETER  = -1, /* EB_ErrorBadParameter */
    SVT_HIP_ERR_NO_RESOURCES   = -2, /* EB_ErrorInsufficientResources */
    SVT_HIP_ERR_DEVICE         = -3, /* HIP runtime failure (message via svt_hip_last_error) */
    SVT_HIP_ERR_UNSUPPORTED    = -4
} svt_hip_status;

typedef struct svt_hip_ctx svt_hip_ctx; /* opaque */

/* ------------------------------------------------------------------------------------------------ */
/* Motion estimation                                                                                  */
/* ------------------------------------------------------------------------------------------------ */

#define SVT_ME_PU_COUNT 85 /* MAX_ME_PU_COUNT, Codec/EbMotionEstimationLcuResults.h:14 */
#define SVT_SB_SIZE 64     /* MAX_SB_SIZE */

/* One padded 8-bit luma plane.  Mirrors the fields of EbPictureBufferDesc that ME reads
 * (Codec/EbPictureBufferDesc.h:27-59): buffer_y, stride_y, origin_x/y (= padding), width, height. */
typedef struct svt_plane {
    const uint8_t *buf;      /* first byte of the padded buffer (top-left of the padding) */
    int32_t        stride;   /* bytes per row */
    int32_t        origin_x; /* left padding  (picture column 0 is at buf[origin_y*stride+origin_x]) */
    int32_t        origin_y; /* top padding */
    int32_t        width;    /* unpadded width  */
    int32_t        height;   /* unpadded height */
} svt_plane;

/* The three planes of an EbPaReferenceObject (Codec/EbReferenceObject.h:39-49):
 * padded input, 1/4 (2x2 point-decimated) and 1/16 (4x4 point-decimated) luma. */
typedef struct svt_pa_picture {
    svt_plane full;
    svt_plane quarter;
    svt_plane sixteenth;
} svt_pa_picture;

/* Fractional search method, Codec/EbDefinitions.h:664-666 */
#define SVT_SUB_SAD_SEARCH 0
#define SVT_FULL_SAD_SEARCH 1
#define SVT_SSD_SEARCH 2

/* Everything motion_estimate_sb reads from PictureParentControlSet / SequenceControlSet / MeContext.
 * Field names follow the reference (Codec/EbMotionEstimationContext.h:372-398,
 * Codec/EbMotionEstimation.c:4584-4631). */
typedef struct svt_me_params {
    /* picture level */
    uint8_t  num_ref_lists;        /* 1 = P_SLICE (list 0 only), 2 = B_SLICE */
    uint8_t  temporal_layer_index;
    uint8_t  hierarchical_levels;  /* selects the HME-L0 multiplier row, Codec/EbDefinitions.h:989-1005 */
    uint8_t  enable_hme_flag;
    uint8_t  enable_hme_level_0_flag;
    uint8_t  enable_hme_level_1_flag;
    uint8_t  enable_hme_level_2_flag;
    uint8_t  cu8x8_mode;           /* 0: refine + bipred 8x8 PUs, 1: full-pel only (Codec/EbDefinitions.h:943) */
    uint8_t  cu16x16_mode;         /* 0: refine + bipred 16x16 PUs */
    uint8_t  same_ref_poc;         /* ref_pic_poc_array[0] == ref_pic_poc_array[1] (EbMotionEstimation.c:4952) */
    uint8_t  rate_control_mode;    /* != 0 -> rcme_distortion is produced */
    /* MeContext multi-mode signals */
    uint8_t  fractional_search_method; /* SVT_SUB_SAD_SEARCH / SVT_FULL_SAD_SEARCH / SVT_SSD_SEARCH */
    uint8_t  fractional_search_model;  /* 0 all, 1 selective (su_pel_enable), 2 off */
    uint8_t  fractional_search64x64;
    uint8_t  single_hme_quadrant;
    /* full-pel search area */
    uint8_t  search_area_width;
    uint8_t  search_area_height;
    /* HME */
    uint16_t number_hme_search_region_in_width;  /* 1 or 2 */
    uint16_t number_hme_search_region_in_height; /* 1 or 2 */
    uint16_t hme_level0_total_search_area_width;
    uint16_t hme_level0_total_search_area_height;
    uint16_t hme_level0_search_area_in_width_array[2];
    uint16_t hme_level0_search_area_in_height_array[2];
    uint16_t hme_level1_search_area_in_width_array[2];
    uint16_t hme_level1_search_area_in_height_array[2];
    uint16_t hme_level2_search_area_in_width_array[2];
    uint16_t hme_level2_search_area_in_height_array[2];
} svt_me_params;

/* Result record per (SB, PU); same field order and size (40 bytes) as MeCuResults
 * (Codec/EbMotionEstimationLcuResults.h:39-56).  The reference's 2-bit `direction` bit-field is
 * widened to a full uint32 here (value 0 = UNI_PRED_LIST_0, 1 = UNI_PRED_LIST_1, 2 = BI_PRED);
 * INTEGRATION.md shows the field-wise copy into MeCuResults. */
typedef struct svt_me_dist_dir {
    uint32_t distortion;
    uint32_t direction;
} svt_me_dist_dir;

typedef struct svt_me_pu_result {
    int16_t         x_mv_l0, y_mv_l0, x_mv_l1, y_mv_l1; /* quarter-pel units */
    svt_me_dist_dir distortion_direction[3];
    uint8_t         total_me_candidate_index;
    uint8_t         pad_[7];
} svt_me_pu_result;

/* Geometry helper: number of SBs of a picture (ceil(w/64)*ceil(h/64)). */
int32_t svt_hip_sb_count(int32_t pic_width, int32_t pic_height);

/* INPUT_SIZE_576p_RANGE_OR_LOWER / 1080i / 1080p / 4K_RANGE = 0..3 of a picture size, by luma sample count
 * (eb_vp9_derive_input_resolution, Codec/EbSequenceControlSet.c:489-499).  It is also the index of the non-moving
 * threshold shift that svt_hip_me_zz_sad_device takes. */
int32_t svt_hip_input_resolution(int32_t pic_width, int32_t pic_height);

/* What the reference's parameter derivation reads from the sequence / picture control sets. */
typedef struct svt_me_picture_config {
    int32_t pic_width, pic_height;   /* luma_width / luma_height */
    int32_t enc_mode;                /* 0..12 */
    int32_t tune;                    /* 0 SQ, 1 OQ, 2 VMAF (Codec/EbDefinitions.h:658-660) */
    int32_t frame_rate;              /* static_config.frame_rate >> 16 (frames per second) */
    int32_t num_ref_lists;           /* 1 = P picture, 2 = B picture */
    int32_t temporal_layer_index;
    int32_t hierarchical_levels;
    int32_t is_used_as_reference;    /* is_used_as_reference_flag */
    int32_t same_ref_poc;            /* both lists hold the same picture */
    int32_t rate_control_mode;
} svt_me_picture_config;

/* Fill `p` exactly as the reference derives the ME signals of a picture with use_default_me_hme = 1, for every tune,
 * enc_mode and picture size it accepts: eb_vp9_signal_derivation_pre_analysis_* (HME enables,
 * Codec/EbResourceCoordinationProcess.c:291-460), eb_vp9_signal_derivation_multi_processes_* (use_subpel_flag, cu8x8_mode,
 * Codec/EbPictureDecisionProcess.c:682-925), eb_vp9_set_me_hme_params_* + eb_vp9_signal_derivation_me_kernel_*
 * (Codec/EbMotionEstimationProcess.c:55-324, 541-720). */
int32_t svt_hip_me_params_derive(svt_me_params *p, const svt_me_picture_config *cfg);

/* Shorthand for the reference's default random-access structure at 60 frames/s: every temporal layer but the deepest is
 * used as reference (is_used_as_reference = temporal_layer_index < hierarchical_levels). */
int32_t svt_hip_me_params_preset(svt_me_params *p, int32_t pic_width, int32_t pic_height, int32_t enc_mode,
                                 int32_t tune, int32_t num_ref_lists, int32_t temporal_layer_index,
                                 int32_t hierarchical_levels);

/* diagnostic: which compiled instance of the ME kernel serves a parameter set -- 0 the generic one, > 0 an instance whose search
 * parameters are compile-time constants (one per BASELINE configuration; identical results by construction) */
int32_t svt_hip_me_kernel_instance(const svt_me_params *p);
/* diagnostic: the instance the last ME launch on ctx actually ran -- the index above, + 100 when the launch was served by the
 * driver for single-region level-0 HME presets (csrc/me_fast.h: whole SB columns, level-0 areas up to 256 x 256), + 200 when it was the pair of
 * launches of the compact LDS layout (csrc/me_layout.h: the 64 x 64-area presets at two workgroups per CU) */
int32_t svt_hip_me_last_instance(const svt_hip_ctx *ctx);
/* diagnostic (host only): LDS bytes of a workgroup of the ME kernel for a parameter set -- compact = 0: the plain layout; 1: the compact one (search-area
 * widths that are multiples of 8; the launcher takes it where it buys a workgroup per CU: 160 KB per CU in granules of 1 280 bytes).  Negative: no such layout. */
int32_t svt_hip_me_lds_bytes(const svt_me_params *p, int32_t compact);
/* Deployment knob of the intra encode pass (svt_hip_encdec_intra_device, and the intra blocks of inter pictures) launched on ctx: at most n
 * one-wave workgroups (0 = the default: one per compute unit, the lowest latency for a key frame alone -- 6.5 ms at 2160p).  A pass that runs
 * BESIDE other work of the device (a key frame of the next GOP beside the current one) leaves more of it to that work with fewer: every CU
 * that hosts one of its waves has registers for one motion-estimation workgroup less (128: 7.9 ms alone, +2 % for the pipelined step of
 * bench.py).  The environment's SVT_HIP_INTRA_WGS sets the process-wide default. */
int32_t svt_hip_ctx_set_intra_workgroups(svt_hip_ctx *ctx, int32_t n);
/* submits an empty kernel to the context's stream and waits: the stream's hardware queue exists afterwards (the runtime creates it at the first
 * submission, 1-3 ms on the submitting thread).  For hosts that time a stream from its first picture (the encoder library's init does this). */
int32_t svt_hip_ctx_warm(svt_hip_ctx *ctx);
/* the same with a kernel that has a private segment of (at least) scratch_bytes per lane, up to 1024: the queue's scratch memory is sized too (the
 * intra pass's kernel needs 752 bytes per lane -- 4 ms on the thread that submits it first; the 32x32 transform 130) */
int32_t svt_hip_ctx_warm_scratch(svt_hip_ctx *ctx, int32_t scratch_bytes);
/* 1 when two parameter sets may share one launch of svt_hip_me_batch_layers_device: equal in every field but num_ref_lists,
 * temporal_layer_index, hierarchical_levels and same_ref_poc (compared field by field: the record has padding) */
int32_t svt_hip_me_params_same_launch(const svt_me_params *a, const svt_me_params *b);

/* ------------------------------------------------------------------------------------------------ */
/* context                                                                                            */
/* ------------------------------------------------------------------------------------------------ */
int32_t svt_hip_ctx_create(svt_hip_ctx **ctx, int32_t device_ordinal);
/* Same, but all work is enqueued on a caller-provided hipStream_t (passed as void*). */
int32_t svt_hip_ctx_create_on_stream(svt_hip_ctx **ctx, int32_t device_ordinal, void *hip_stream);
/* A context whose own stream is restricted to a set of compute units (hipExtStreamCreateWithCUMask; bit i of the mask
 * words = CU i in the driver's enumeration).  Stages of a pipeline that run concurrently can be given disjoint CU sets
 * so that each CU executes one kernel (its LDS, instruction cache and issue slots are not shared between stages). */
int32_t svt_hip_ctx_create_cu_mask(svt_hip_ctx **ctx, int32_t device, const uint32_t *cu_mask, int32_t mask_words);
/* the context's HIP stream (hipStream_t), e.g. to record events on it or to make other streams wait for it */
void *svt_hip_ctx_stream(svt_hip_ctx *ctx);
void    svt_hip_ctx_destroy(svt_hip_ctx *ctx);
int32_t svt_hip_ctx_synchronize(svt_hip_ctx *ctx);
const char *svt_hip_last_error(void);
/* Timing of the kernels launched by the most recent *_device call on this context, measured with
 * hipEvents recorded on the context's stream (ms).  Valid after svt_hip_ctx_synchronize(). */
float   svt_hip_last_kernel_ms(svt_hip_ctx *ctx);

/* Device memory for C hosts (the *_device entry points take HBM pointers): plain allocations on the context's device, and
 * copies ordered on the context's stream.  svt_hip_mem_upload_2d returns after the host rows have been consumed (the caller
 * may reuse them), svt_hip_mem_download after the data has arrived. */
int32_t svt_hip_mem_alloc(svt_hip_ctx *ctx, size_t bytes, void **d_ptr);
void    svt_hip_mem_free(svt_hip_ctx *ctx, void *d_ptr);
int32_t svt_hip_mem_upload_2d(svt_hip_ctx *ctx, void *d_dst, size_t dst_stride, const void *src, size_t src_stride, size_t width_bytes,
                              size_t rows);
int32_t svt_hip_mem_download(svt_hip_ctx *ctx, void *dst, const void *d_src, size_t bytes);
/* Asynchronous form of svt_hip_mem_upload_2d for a host that must not wait for the device: the rows are copied into a pinned
 * staging buffer owned by the context before the call returns (= the copy eb_vp9_svt_enc_send_picture makes of the caller's
 * picture, Codec/EbEncHandle.c:2743-2796: the caller may reuse its buffer at once), the host-to-device copy is enqueued on the
 * context's stream.  The call blocks only when all staging buffers (4) still hold copies the device has not consumed. */
int32_t svt_hip_mem_upload_2d_async(svt_hip_ctx *ctx, void *d_dst, size_t dst_stride, const void *src, size_t src_stride, size_t width_bytes,
                                    size_t rows);
/* Up to 4 planes of one picture through ONE staging slot (same contract as svt_hip_mem_upload_2d_async for each plane): destinations
 * that lie back to back on the device, tight (dst_stride == width_bytes, d_dst[i + 1] == the end of plane i: Y | Cb | Cr in one buffer),
 * travel as one host-to-device copy -- a third of the stream operations of three separate uploads (copy_frame_buffer copies the three
 * planes of a picture in one call too, Codec/EbEncHandle.c:2743-2796). */
int32_t svt_hip_mem_upload_planes_async(svt_hip_ctx *ctx, int32_t n_planes, void *const *d_dst, const size_t *dst_stride, const void *const *src,
                                        const size_t *src_stride, const size_t *width_bytes, const size_t *rows);
/* The same upload WITHOUT the staging copy, for a host whose rows lie in memory that stays allocated while the library is in use: a
 * range of host memory is page-locked (hipHostRegister) the first time it is seen, so a host that sends from a fixed pool of buffers
 * pays for that once and is read by the DMA engines directly from then on (57 GB/s on the MI355X box against ~27 GB/s through the
 * staging copy); ranges that cannot be registered, and more than 1024 distinct ranges, go through the staging path.  The copy is
 * asynchronous: the rows must stay valid until svt_hip_mem_upload_wait(ctx) has returned (it waits for every direct upload enqueued
 * on ctx -- not for the rest of the stream).  svt_hip_host_unregister_all unlocks everything (no upload may be in flight).  Locked
 * ranges must not be freed, nor passed in part to other host <-> device copies of the process, before they are unlocked. */
int32_t svt_hip_mem_upload_2d_direct(svt_hip_ctx *ctx, void *d_dst, size_t dst_stride, const void *src, size_t src_stride, size_t width_bytes,
                                     size_t rows);
int32_t svt_hip_mem_upload_wait(svt_hip_ctx *ctx);
void    svt_hip_host_unregister_all(void);
/* The registry's life follows its users: a host retains it while it uploads directly and releases it when all its uploads have completed;
 * the LAST release unlocks every range (memory freed afterwards, or re-allocated at the same address, is never read through a stale
 * registration).  A range whose registration failed is not remembered: the next upload from it tries again.  The encoder library
 * (libSvtVp9Enc.so, SVT_HIP_REGISTER_INPUT=1) retains in eb_vp9_init_encoder and releases in eb_vp9_deinit_encoder. */
void    svt_hip_host_registry_retain(void);
void    svt_hip_host_registry_release(void);
/* Completion markers: svt_hip_ctx_marker_record notes "everything enqueued on the context's stream so far" and returns a marker;
 * svt_hip_ctx_marker_query returns 1 when that work has completed, 0 while it is pending (never blocks), negative on error;
 * svt_hip_ctx_marker_wait blocks until it has.  This is what lets eb_vp9_svt_get_packet poll without blocking and block only when
 * the application says it has sent its last picture (Codec/EbEncHandle.c:2880-2915). */
int32_t svt_hip_ctx_marker_record(svt_hip_ctx *ctx, uint64_t *marker);
int32_t svt_hip_ctx_marker_query(svt_hip_ctx *ctx, uint64_t marker);
int32_t svt_hip_ctx_marker_wait(svt_hip_ctx *ctx, uint64_t marker);
int32_t svt_hip_mem_set(svt_hip_ctx *ctx, void *d_dst, int32_t value, size_t bytes);

/* ------------------------------------------------------------------------------------------------ */
/* ME entry points                                                                                    */
/* ------------------------------------------------------------------------------------------------ */

/* Device-resident form (the one that is benchmarked): every svt_plane.buf in cur/ref0/ref1 is a
 * device pointer, d_results is a device array [n_sb][85].  ref1 may be NULL when num_ref_lists==1.
 * d_rcme_distortion (uint32 per SB) may be NULL.  Asynchronous on the context's stream. */
int32_t svt_hip_me_picture_device(svt_hip_ctx *ctx, const svt_pa_picture *cur, const svt_pa_picture *ref0,
                                  const svt_pa_picture *ref1, const svt_me_params *params,
                                  svt_me_pu_result *d_results, uint32_t *d_rcme_distortion);

/* Batched device-resident form: ME of n_pics independent pictures in ONE launch (ME reads source
 * pictures only -- Codec/EbMotionEstimation.c:4638-4642 -- so all pictures of a mini-GOP are
 * independent).  Arrays of n_pics descriptors (host memory); d_results[i] / d_rcme[i] device arrays. */
int32_t svt_hip_me_batch_device(svt_hip_ctx *ctx, int32_t n_pics, const svt_pa_picture *cur,
                                const svt_pa_picture *ref0, const svt_pa_picture *ref1,
                                const svt_me_params *params, svt_me_pu_result *const *d_results,
                                uint32_t *const *d_rcme_distortion);
/* The same with one parameter set PER PICTURE (params[i] for picture i): the pictures of a mini-GOP differ in
 * num_ref_lists, temporal_layer_index, hierarchical_levels and same_ref_poc only (everything else follows from the
 * configuration: Codec/EbMotionEstimationProcess.c:541-720), so one launch serves all its temporal layers -- no launch
 * tails between the layers.  Returns SVT_HIP_ERR_BAD_PARAMETER when the sets differ in any other field. */
int32_t svt_hip_me_batch_layers_device(svt_hip_ctx *ctx, int32_t n_pics, const svt_pa_picture *cur,
                                       const svt_pa_picture *ref0, const svt_pa_picture *ref1,
                                       const svt_me_params *params, svt_me_pu_result *const *d_results,
                                       uint32_t *const *d_rcme_distortion);

/* Host-pointer convenience form (what the reference's ME thread would call): uploads the planes,
 * runs svt_hip_me_picture_device, downloads results[n_sb][85], synchronous. */
int32_t svt_hip_me_picture(svt_hip_ctx *ctx, const svt_pa_picture *cur, const svt_pa_picture *ref0,
                           const svt_pa_picture *ref1, const svt_me_params *params, svt_me_pu_result *results,
                           uint32_t *rcme_distortion);

/* Stand-alone HME level-0 search = eb_vp9_sad_loop_kernel (C_DEFAULT/EbComputeSAD_C.c:132-169)
 * batched over n independent (block, window) problems; device pointers. Exposed for unit parity tests.
 * Each problem: block w x h (rows already at src_stride), window search_w x search_h; ref row step
 * for the block rows = ref_stride, for successive search rows = ref_stride_raw. */
typedef struct svt_sad_loop_job {
    uint64_t src_off;   /* byte offset into d_src */
    uint64_t ref_off;   /* byte offset into d_ref */
    int32_t  src_stride, ref_stride, ref_stride_raw;
    int32_t  width, height;       /* block */
    int32_t  search_w, search_h;  /* window */
} svt_sad_loop_job;
typedef struct svt_sad_loop_result {
    uint32_t best_sad;
    int16_t  x, y;
} svt_sad_loop_result;
int32_t svt_hip_sad_loop_batch_device(svt_hip_ctx *ctx, const uint8_t *d_src, const uint8_t *d_ref,
                                      const svt_sad_loop_job *d_jobs, int32_t n_jobs,
                                      svt_sad_loop_result *d_out);

/* Per-SB side outputs of the ME stage (row M12 of the scope table).
 *
 * svt_hip_me_zz_sad_device = compute_zz_sad (Codec/EbMotionEstimationProcess.c:431-534): for every complete SB the
 * 16x16 SAD between the current picture's 1/16 plane and the 4x4 point-decimated collocated SB of the PREVIOUS
 * picture's input (eb_vp9_decimation_2d, Codec/EbPictureAnalysisProcess.c:102-122); incomplete SBs get 0xffffffff.
 * d_non_moving_index[sb] = NON_MOVING_SCORE_{0,1,2,3} = 0/10/20/30 from the thresholds 2,4,8 x (16*16) >>
 * non_moving_th_shift[input_resolution] (:353; input_resolution 0..3 = <=576p, 720p, 1080p, 2160p) -- the value the
 * reference stores into the previous picture's non_moving_index_array.  Planes are device resident. */
int32_t svt_hip_me_zz_sad_device(svt_hip_ctx *ctx, const svt_plane *cur_sixteenth, const svt_plane *prev_input,
                                 int32_t input_resolution, uint32_t *d_zz_sad, uint8_t *d_non_moving_index);

/* Host-side: eb_vp9_derive_similar_collocated_flag (Codec/EbMotionEstimationProcess.c:747-783) for n SBs: compares the
 * 64x64 mean / variance of each SB (picture analysis outputs) with the list-0 reference's.  similar[sb] is only set
 * when the picture is used as a reference, similar_all_layers[sb] always. */
void svt_hip_me_similar_collocated(const uint8_t *cur_mean, const uint16_t *cur_var, const uint8_t *ref_mean,
                                   const uint16_t *ref_var, int32_t n_sb, int32_t is_i_slice,
                                   int32_t is_used_as_reference, uint8_t *similar, uint8_t *similar_all_layers);

/* The rest of the ME kernel process's per-SB bookkeeping (scope row M12), on the device so that neither the 6.9 MB result
 * array nor the picture-analysis statistics have to come back to the host for it:
 *   stationary_edge_over_update_over_time_sb_part1 / _part2 (Codec/EbMotionEstimationProcess.c:785-869): the logo /
 *       stationary-edge flags of every SB from the 64x64 list-0 MV and distortion of its best ME candidate (d_results[sb][0])
 *       and the variance of its four 32x32 blocks (d_var[sb * 85 + 1..4], svt_hip_pa_mean_variance_device's layout);
 *       which SBs can hold a logo is eb_vp9_sb_params_init's potential_logo_sb rule (Codec/EbSequenceControlSet.c:332-413);
 *   the rate-control SAD-interval indices and histograms (:1103-1237): inter index from rcme_distortion[sb] (the ME entry
 *       points' d_rcme_distortion), intra index from the 64x64 variance, one count per complete SB in each histogram.
 * part2 only runs when run_part2 != 0 (= !end_of_sequence_flag && look_ahead_distance != 0); without it check2 /
 * low_dist_logo are written as 0.  The histograms and the full-SB count are ADDED to (the reference zeroes them per picture). */
typedef struct svt_me_sb_stats_params {
    int32_t pic_width, pic_height;
    int32_t input_resolution;      /* svt_hip_input_resolution() */
    int32_t temporal_layer_index;
    int32_t slice_type;            /* 0 B_SLICE, 1 P_SLICE, 2 I_SLICE (Codec/EbDefinitions.h EB_SLICE) */
    int32_t run_part2;
    int32_t rate_control_mode;     /* 0: no indices / histograms (static_config.rate_control_mode) */
} svt_me_sb_stats_params;
typedef struct svt_me_sb_stats {
    uint8_t  check1_for_logo_stationary_edge_over_time_flag;
    uint8_t  pm_check1_for_logo_stationary_edge_over_time_flag;
    uint8_t  check2_for_logo_stationary_edge_over_time_flag;
    uint8_t  low_dist_logo;
    uint16_t inter_sad_interval_index;
    uint16_t intra_sad_interval_index;
} svt_me_sb_stats;                 /* 8 bytes */
#define SVT_SAD_INTERVALS 128      /* NUMBER_OF_SAD_INTERVALS, Codec/EbRateControlTables.h:19 */
/* d_results may be NULL for an I picture, d_rcme_distortion when rate_control_mode == 0 or for an I picture.
 * d_hist: [0..127] me_distortion_histogram, [128..255] ois_distortion_histogram; d_full_sb_count: one uint32. */
int32_t svt_hip_me_sb_stats_device(svt_hip_ctx *ctx, const svt_me_sb_stats_params *params, const svt_me_pu_result *d_results,
                                   const uint16_t *d_var, const uint32_t *d_rcme_distortion, svt_me_sb_stats *d_stats,
                                   uint32_t *d_hist, uint32_t *d_full_sb_count);

/* ------------------------------------------------------------------------------------------------ */
/* Picture-analysis pre-ME stage ("next" row f-1 of the scope table)                                  */
/* ------------------------------------------------------------------------------------------------ */

/* Builds the three planes of an EbPaReferenceObject from a picture's W x H luma, all on the device:
 *   full      = copy + edge replication by origin_x/origin_y samples (eb_vp9_generate_padding, Codec/EbMcp.c:17-58,
 *               called by pad_picture_to_multiple_of_sb_dimensions, Codec/EbPictureAnalysisProcess.c:5010-5020)
 *   quarter   = 2x2 point decimation (eb_vp9_decimation_2d, :102-122, step 2) + edge replication, only if make_quarter
 *   sixteenth = 4x4 point decimation (step 4) + edge replication       (decimate_input_picture, :5025-5088)
 * `out[i]` describes the destination planes of picture i (device buffers; buf, stride, origin, width, height as the
 * ME entry points expect them: quarter = W/2 x H/2, sixteenth = W/4 x H/4).  W and H must be multiples of 8. */
int32_t svt_hip_pa_prepare_batch_device(svt_hip_ctx *ctx, int32_t n_pics, const uint8_t *const *d_luma,
                                        const int32_t *luma_stride, const svt_pa_picture *out, int32_t make_quarter);

/* 8x8 .. 64x64 block mean and variance of every SB of a padded luma plane = compute_block_mean_compute_variance
 * (Codec/EbPictureAnalysisProcess.c:2115-3356; 8x8 sums on rows 0,2,4,6 with the SSE2 kernels the reference calls at
 * `-asm 0`, ASM_SSE2/EbComputeMean_Intrinsic_SSE2.c:10-53; larger blocks are >>2 averages of their four children).
 * d_mean[sb*85 + pu] = (uint8_t)(mean >> 8), d_var[sb*85 + pu] = (uint16_t)((mean_of_squares - mean^2) >> 16), pu in
 * raster order per size: 0 = 64x64, 1-4 = 32x32, 5-20 = 16x16, 21-84 = 8x8 (ME_TIER_ZERO_PU_*,
 * Codec/EbMotionEstimationContext.h:45-131).  SBs at the right/bottom border read the replicated padding. */
int32_t svt_hip_pa_mean_variance_device(svt_hip_ctx *ctx, const svt_plane *full, uint8_t *d_mean, uint16_t *d_var);

/* ------------------------------------------------------------------------------------------------ */
/* Transform / quantisation                                                                           */
/* ------------------------------------------------------------------------------------------------ */

/* TX sizes / types follow VPX/vp9_enums.h (TX_4X4..TX_32X32; DCT_DCT, ADST_DCT, DCT_ADST, ADST_ADST). */
#define SVT_TX_4X4 0
#define SVT_TX_8X8 1
#define SVT_TX_16X16 2
#define SVT_TX_32X32 3
#define SVT_DCT_DCT 0
#define SVT_ADST_DCT 1
#define SVT_DCT_ADST 2
#define SVT_ADST_ADST 3

/* Quantiser tables of one (qindex, plane): the [0]=DC,[1]=AC pairs eb_vp9_init_quantizer produces
 * (VPX/vp9_quantize.c:206-265); passed to eb_vp9_quantize_b[_32x32] (VPX/quantize.c:112-254). */
typedef struct svt_quant_tables {
    int16_t zbin[2], round[2], quant[2], quant_shift[2], dequant[2];
} svt_quant_tables;

/* Host-side: the tables of one (q index, plane) as eb_vp9_init_quantizer derives them (sharpness 0): q_index 0..255;
 * y_dc_step = eb_vp9_dc_quant(q_index, 0) (it selects the zero-bin factor, get_qzbin_factor :192-204); dc_step / ac_step =
 * eb_vp9_dc_quant / eb_vp9_ac_quant with the plane's deltas (VPX/vp9_quant_common.c). */
int32_t svt_hip_quant_tables_init(int32_t q_index, int32_t y_dc_step, int32_t dc_step, int32_t ac_step, svt_quant_tables *out);

/* One transform block of perform_coding_loop (Codec/EbEncDecProcess.c:365-587). */
typedef struct svt_tq_block {
    uint32_t src_off;    /* byte offset of the block's top-left in the source plane  */
    uint32_t pred_off;   /* byte offset in the prediction plane                       */
    uint32_t recon_off;  /* byte offset in the recon plane (written when do_recon)    */
    uint32_t coeff_off;  /* element offset into qcoeff / dqcoeff (n*n contiguous); multiple of 8 */
    uint32_t iscan_off;  /* element offset of this block's iscan table inside the iscan array */
    uint16_t src_stride, pred_stride, recon_stride;
    uint8_t  tx_size;    /* SVT_TX_* */
    uint8_t  tx_type;    /* SVT_DCT_DCT.. (ADST only for <=16x16) */
    uint8_t  qtab;       /* index into the quant-table array */
    uint8_t  do_recon;   /* run inverse transform + add into recon */
    uint8_t  partial32;  /* 32x32 only: use eb_vpx_partial_fdct32x32 (low 16x16 kept, rest zero) */
    uint8_t  pad_[1];    /* svt_hip_tq_rd_batch_device only: SVT_TQ_RATE_INFO(ctx, plane_type, is_inter), the block's rate inputs */
} svt_tq_block;          /* 32 bytes */
/* rate inputs of a block for the fused distortion + rate entry: ctx = combine_entropy_contexts(left, above) 0..2
 * (Codec/EbEncDecProcess.c:734), plane_type = get_plane_type (0 luma, 1 chroma), is_inter = is_inter_block(mi) */
#define SVT_TQ_RATE_INFO(ctx, plane_type, is_inter) ((uint8_t)(((ctx) & 3) | (((plane_type) & 1) << 2) | (((is_inter) & 1) << 3)))

/* Scan tables: the caller passes the reference's own iscan tables (scan_order->iscan of
 * eb_vp9_scan_orders[tx_size][tx_type] / eb_vp9_default_scan_orders[TX_32X32], VPX/vp9_scan.c) concatenated
 * in one int16 array; every block names its table by element offset (iscan_off).  iscan[rc] = position of
 * raster coefficient rc in scan order, which is all the quantiser needs to produce eob.
 *
 * residual = src - pred (eb_vp9_residual_kernel, C_DEFAULT/EbPictureOperators_C.c:204-223)
 * -> forward DCT/ADST (VPX/fwd_txfm.c, VPX/vp9_dct.c) -> eb_vp9_quantize_b[_32x32]
 * -> if do_recon: recon = pred; inverse transform of dqcoeff added into recon
 *    (eb_vp9_idct*_add / eb_vp9_iht*_add, VPX/vp9_idct.c:111-189).
 * Outputs: qcoeff, dqcoeff (int16, raster within block), eob per block.  Device pointers.
 * d_blocks must be GROUPED by transform size in the order 4x4, 8x8, 16x16, 32x32 with size_count[s] blocks of
 * size s (size_count is a host array), and do_recon must be uniform within a size group (the encode pass
 * reconstructs every block, mode decision none).  d_qcoeff / d_dqcoeff are 16-byte aligned and every coeff_off is a
 * multiple of 8 (rows are stored as 16-byte vectors; the reference's per-SB coefficient buffers advance by whole
 * blocks of >= 16 coefficients, so its offsets always are). */
int32_t svt_hip_tq_batch_device(svt_hip_ctx *ctx, const uint8_t *d_src, const uint8_t *d_pred, uint8_t *d_recon,
                                const svt_tq_block *d_blocks, const int32_t size_count[4],
                                const svt_quant_tables *d_qtabs, const int16_t *d_iscan, int16_t *d_qcoeff,
                                int16_t *d_dqcoeff, uint16_t *d_eob);

/* Host-pointer convenience form. plane_bytes = size of each of src/pred/recon; coeff_count = total coeffs. */
/* Same, additionally producing the coefficient-domain distortion pair of every block as the reference's
 * full_distortion_kernel32bit does right after the transform/quantisation in mode decision
 * (C_DEFAULT/EbPictureOperators_C.c:288-311; function table Codec/EbPictureOperators.h:240):
 * d_dist[2*b] = sum (coeff - dqcoeff)^2, d_dist[2*b+1] = sum coeff^2 over the NxN block, with the reference's
 * int16/uint32 wrap-around.  (The _intra and _eob_zero variants of the reference return one of the two values twice.) */
int32_t svt_hip_tq_batch_dist_device(svt_hip_ctx *ctx, const uint8_t *d_src, const uint8_t *d_pred, uint8_t *d_recon,
                                     const svt_tq_block *d_blocks, const int32_t size_count[4],
                                     const svt_quant_tables *d_qtabs, const int16_t *d_iscan, int16_t *d_qcoeff,
                                     int16_t *d_dqcoeff, uint16_t *d_eob, uint64_t *d_dist);

/* Forward declaration (defined with the coefficient rate section below). */
struct svt_rate_tables;

/* perform_dist_rate_calc in one pass (Codec/EbEncDecProcess.c:700-745): svt_hip_tq_batch_dist_device plus, per block,
 * d_bits[b] = coeff_rate_estimate(...) (Codec/EbRateDistortionCost.c:55-172, use_fast_coef_costing = 0) of the coefficients the
 * quantiser has just produced -- computed from registers / LDS behind the quantiser, no second pass over d_qcoeff.  The
 * block's rate inputs travel in svt_tq_block.pad_[0] (SVT_TQ_RATE_INFO).  d_tables / d_scan as for
 * svt_hip_coeff_rate_batch_device, with d_scan in the CANONICAL layout: for tx_size 0..3, for tx_type 0..3 the table
 * {scan[n], neighbors[2 (n + 1)]} of eb_vp9_scan_orders[tx_size][tx_type] (for 32x32 the four slots all hold the
 * default order); a block's table is found from its tx_size / tx_type.  4x4 blocks are walked with the three VP9 4x4 scan
 * orders compiled into the kernel (svt_hip_rate_scan4x4_table returns them; d_scan must hold the same, normative, ones). */
int32_t svt_hip_tq_rd_batch_device(svt_hip_ctx *ctx, const uint8_t *d_src, const uint8_t *d_pred, uint8_t *d_recon,
                                   const svt_tq_block *d_blocks, const int32_t size_count[4],
                                   const svt_quant_tables *d_qtabs, const int16_t *d_iscan, int16_t *d_qcoeff,
                                   int16_t *d_dqcoeff, uint16_t *d_eob, uint64_t *d_dist,
                                   const struct svt_rate_tables *d_tables, const int16_t *d_scan, int32_t *d_bits);
/* The same for a batch whose blocks reconstruct into up to 8 different buffers: block b writes into recon_set[i] + recon_off with
 * i = bits 4-6 of pad_[0] (SVT_TQ_RECON_SET(i), OR-ed with the rate info).  recon_set is a HOST array of n_set device pointers; a
 * block whose i >= n_set is transformed and quantised like any other but its reconstruction is written nowhere.
 * In the reference every picture reconstructs into its own reference-picture buffer (Codec/EbEncDecProcess.c:5658-5674); a step of
 * a picture-level pipeline codes pictures of several mini-GOPs whose buffers need not lie within one 32-bit offset of each other:
 * this entry serves them with one launch per transform size.  d_bits == NULL: no rate (then d_tables / d_scan are ignored). */
#define SVT_TQ_RECON_SET(i) ((uint8_t)(((i) & 7) << 4))
int32_t svt_hip_tq_rd_batch_multi_device(svt_hip_ctx *ctx, const uint8_t *d_src, const uint8_t *d_pred, uint8_t *const *recon_set, int32_t n_set,
                                         const svt_tq_block *d_blocks, const int32_t size_count[4], const svt_quant_tables *d_qtabs,
                                         const int16_t *d_iscan, int16_t *d_qcoeff, int16_t *d_dqcoeff, uint16_t *d_eob, uint64_t *d_dist,
                                         const struct svt_rate_tables *d_tables, const int16_t *d_scan, int32_t *d_bits);
/* host: the 4x4 scan order (16 entries) and neighbour pairs (32 entries) the fused rate pass uses for tx_type 0..3 */
int32_t svt_hip_rate_scan4x4_table(int32_t tx_type, int16_t out[48]);

int32_t svt_hip_tq_batch(svt_hip_ctx *ctx, const uint8_t *src, const uint8_t *pred, uint8_t *recon,
                         size_t plane_bytes, const svt_tq_block *blocks, int32_t n_blocks,
                         const svt_quant_tables *qtabs, int32_t n_qtabs, const int16_t *iscan, size_t iscan_count,
                         int16_t *qcoeff, int16_t *dqcoeff, size_t coeff_count, uint16_t *eob);

/* ------------------------------------------------------------------------------------------------ */
/* In-loop deblocking                                                                                 */
/* ------------------------------------------------------------------------------------------------ */

/* Bit-for-bit the reference's LOOP_FILTER_MASK (VPX/vp9_loopfilter.h; 160 bytes):
 * built on the host by eb_vp9_build_mask_frame (VPX/vp9_loopfilter.c:1548) and uploaded as is. */
typedef struct svt_lf_mask {
    uint64_t left_y[4];   /* per TX size */
    uint64_t above_y[4];
    uint64_t int_4x4_y;
    uint16_t left_uv[4];
    uint16_t above_uv[4];
    uint16_t int_4x4_uv;
    uint8_t  lfl_y[64];
} svt_lf_mask;

/* Per-level thresholds = loop_filter_info_n.lfthr[] (VPX/vp9_loopfilter.h: mblim/lim/hev_thr, each
 * SIMD_WIDTH=16 replicated bytes); only byte 0 of each is meaningful to the C kernels. */
typedef struct svt_lf_thresh {
    uint8_t mblim[64];
    uint8_t lim[64];
    uint8_t hev_thr[64];
} svt_lf_thresh;

/* Compute the thresholds exactly as eb_vp9_loop_filter_init/update_sharpness
 * (VPX/vp9_loopfilter.c:221-262) for a given sharpness level. */
void svt_hip_lf_thresh_init(svt_lf_thresh *t, int32_t sharpness_level);
/* eb_vp9_pick_filter_level's level-from-q rule (VPX/vp9_picklpf.c:37-89), 8-bit. */
int32_t svt_hip_lf_level_from_q(int32_t ac_q, int32_t is_key_frame);

/* The fields of ModeInfo (VPX/vp9_blockd.h) that the mask builder reads, one record per 8x8 unit of the picture
 * (mi_rows x mi_stride grid).  Every 8x8 unit covered by a prediction block carries that block's values, exactly as
 * every entry of cm->mi_grid_visible[] points at the block's ModeInfo in the reference. */
typedef struct svt_lf_mode_info {
    uint8_t sb_type;      /* BLOCK_SIZE: 0 4x4, 1 4x8, 2 8x4, 3 8x8, 4 8x16, 5 16x8, 6 16x16, 7 16x32, 8 32x16, 9 32x32,
                             10 32x64, 11 64x32, 12 64x64 (VPX/vp9_enums.h) */
    uint8_t tx_size;      /* TX_4X4 .. TX_32X32 */
    uint8_t skip;         /* no coefficients coded */
    uint8_t is_inter;     /* ref_frame[0] > INTRA_FRAME */
    uint8_t filter_level; /* get_filter_level(): lf_info.lvl[segment_id][ref_frame[0]][mode_lf_lut[mode]] */
    uint8_t pad_[3];
} svt_lf_mode_info;

/* Host-side: builds the LOOP_FILTER_MASK of every SB from the mode-info grid = eb_vp9_build_mask_frame
 * (VPX/vp9_loopfilter.c:1548-1571 -> eb_vp9_setup_mask :901-1040 / per block eb_vp9_build_mask :1587-1689).
 * lfm[(mi_row >> 3) * lfm_stride + (mi_col >> 3)] is written for every SB (zeroed first, like eb_vp9_setup_mask).
 * The result is the input of svt_hip_lf_frame (eb_vp9_adjust_mask is applied inside the filter, as in the reference). */
int32_t svt_hip_lf_build_masks(const svt_lf_mode_info *mi, int32_t mi_stride, int32_t mi_rows, int32_t mi_cols,
                               svt_lf_mask *lfm, int32_t lfm_stride);

/* 4:2:0 recon picture, planes filtered in place. */
typedef struct svt_yuv_planes {
    uint8_t *y, *u, *v;    /* pointers to picture sample (0,0) of each plane */
    int32_t  y_stride, uv_stride;
    int32_t  width, height; /* luma dimensions (multiples of 8) */
} svt_yuv_planes;

/* Takes the context's per-SB edge-descriptor buffer (1 280 bytes per SB) for launches of up to n_pics pictures now: it grows on demand, but growing
 * waits for the context's stream.  The encoder library calls this when it is initialised. */
int32_t svt_hip_lf_reserve(svt_hip_ctx *ctx, int32_t n_pics, int32_t mi_rows, int32_t mi_cols);

/* = eb_vp9_loop_filter_frame(frame, cm, xd, lfm_base, filter_level, y_only=0, partial=0)
 * (VPX/vp9_loopfilter.c:1521-1546 -> loop_filter_rows :1456).  lfm: one mask per SB in raster order
 * [ceil(mi_rows/8)][lfm_stride].  Device pointers.  mi_rows/mi_cols in 8x8 units. */
int32_t svt_hip_lf_frame_device(svt_hip_ctx *ctx, const svt_yuv_planes *d_recon, const svt_lf_mask *d_lfm,
                                int32_t lfm_stride, const svt_lf_thresh *thr, int32_t mi_rows, int32_t mi_cols,
                                int32_t y_only);
/* Several independent frames in one launch (their SB-row wavefronts interleave and fill the GPU).  Host arrays of
 * n_pics entries; every pointer inside them is a device pointer. */
int32_t svt_hip_lf_batch_device(svt_hip_ctx *ctx, int32_t n_pics, const svt_yuv_planes *d_recon,
                                const svt_lf_mask *const *d_lfm, const int32_t *lfm_stride, const svt_lf_thresh *thr,
                                const int32_t *mi_rows, const int32_t *mi_cols, int32_t y_only);
/* Host-pointer convenience form (planes are tightly described by recon; rows*stride bytes copied). */
int32_t svt_hip_lf_frame(svt_hip_ctx *ctx, const svt_yuv_planes *recon, const svt_lf_mask *lfm, int32_t lfm_stride,
                         const svt_lf_thresh *thr, int32_t mi_rows, int32_t mi_cols, int32_t y_only);

/* ------------------------------------------------------------------------------------------------------------------
 * Inter prediction (8-tap motion compensation) of a whole picture -- SURVEY 8(f) row 2.
 *
 * Replaces the inter branch of prediction_fun_table (Codec/EbEncDecProcess.c:132, called per block and plane at
 * :277 / :3800): inter_prediction (Codec/EbIntraPrediction.c:49-72) -> build_inter_predictors
 * (VPX/vp9_reconinter.c:102-252) -> eb_vp9_clamp_mv_to_umv_border_sb (:72-92) -> inter_predictor
 * (VPX/vp9_reconinter.h:23-28) -> eb_vpx_convolve_copy / eb_vp9_convolve8[_horiz|_vert] and their _avg forms for the
 * second reference of a compound block (VPX/vpx_convolve.c:20-215) with the regular 8-tap kernel
 * (VPX/vp9_filter.c:32-47; the reference hard-wires eb_vp9_filter_kernels[0], vp9_reconinter.c:107-109).
 * Unscaled references only (sf->x_step_q4 == y_step_q4 == 16, the only case the reference's scale setup keeps,
 * VPX/vp9_scale.c:76-84, 126-128).  The prediction written here is the d_pred input of svt_hip_tq_batch_device.
 *
 * The picture is described the way the reference describes it after mode decision: one record per 8x8 unit of the
 * mode-info grid (cm->mi_grid_visible), every unit of a block carrying the block's values.  Blocks are the
 * reference's >= 8x8 partitions (inter_prediction asserts that no sub-8x8 inter block exists, :62-64), aligned to
 * their own size. */
typedef struct svt_mc_mode_info {
    int16_t mv_row[2], mv_col[2]; /* mi->mv[ref].as_mv, 1/8 luma sample (MV_PRECISION_Q3) */
    int8_t  ref_list[2];          /* reference list of ref 0 / ref 1: 0 = REF_LIST_0 (LAST_FRAME), 1 = REF_LIST_1, -1 = none;
                                     ref_list[0] < 0: not an inter block, nothing is written for this unit;
                                     ref_list[1] >= 0: compound (has_second_ref) */
    uint8_t bw8, bh8;             /* block width / height in 8x8 units (1, 2, 4, 8): num_8x8_blocks_wide/high_lookup[sb_type] */
} svt_mc_mode_info;               /* 12 bytes */

/* One picture: d_mi[mi_row * mi_stride + mi_col]; ref[l] = ref_pic_list[l] (planes need the reference's padding of
 * at least 64 + 16 samples around the luma picture and half of it around chroma, Codec/EbEncHandle.c:968-970: a
 * clamped MV reaches bw + 4 samples beyond the edge and the filter 4 more); pred = the prediction picture (sample
 * (0,0) pointers).  use_subpel = context_ptr->use_subpel_flag (0: the MVs are rounded to full samples the way
 * vp9_reconinter.c:175-188 does).  All pointers are device pointers. */
typedef struct svt_mc_picture {
    const svt_mc_mode_info *d_mi;
    int32_t        mi_stride, mi_rows, mi_cols;
    svt_yuv_planes ref[2];
    svt_yuv_planes pred;
    int32_t        use_subpel;
} svt_mc_picture;

/* n_pics independent pictures in one launch (host array of descriptors holding device pointers). */
int32_t svt_hip_inter_pred_batch_device(svt_hip_ctx *ctx, int32_t n_pics, const svt_mc_picture *pics);
/* Host-pointer convenience form for one picture: plane buffers are given WITH their padding (buf = first byte of the
 * padded plane, org_x/org_y = padding of the luma plane; chroma planes have half of it); the three prediction planes
 * are tight (width x height, width/2 x height/2) and are read back. */
typedef struct svt_mc_host_ref {
    const uint8_t *y, *u, *v;        /* padded planes */
    int32_t        y_stride, uv_stride;
    int32_t        org_x, org_y;     /* luma padding (even) */
} svt_mc_host_ref;
int32_t svt_hip_inter_pred_frame(svt_hip_ctx *ctx, const svt_mc_mode_info *mi, int32_t mi_stride, int32_t mi_rows,
                                 int32_t mi_cols, const svt_mc_host_ref ref[2], int32_t use_subpel, uint8_t *pred_y,
                                 uint8_t *pred_u, uint8_t *pred_v);

/* ------------------------------------------------------------------------------------------------------------------
 * Deblocked reconstruction -> reference picture: the step that closes the loop of the path.
 *
 * Replaces pad_ref_and_set_flags (Codec/EbEncDecProcess.c:4822-4851, called after the loop filter at :5696 for every picture
 * with is_used_as_reference_flag) -> eb_vp9_generate_padding (Codec/EbMcp.c:17-58) on Y with (pad_x, pad_y) and on Cb / Cr
 * with (pad_x >> 1, pad_y >> 1): the picture's edge samples are replicated into the border of its buffer, in place (in the
 * reference a reference picture's reconstruction buffer IS the reference picture, allocated with 64 + 16 samples of border,
 * Codec/EbEncHandle.c:968-971).  pics[i] gives the sample (0,0) pointers and strides of picture i (device memory; the
 * border of pad_y rows / pad_x columns around each plane must belong to the buffer: stride >= width + 2 pad_x).  The
 * padded planes are what svt_mc_picture.ref[] expects and what svt_hip_ref_handoff_device ships.  Host array of
 * descriptors; asynchronous on the context's stream. */
int32_t svt_hip_ref_pad_batch_device(svt_hip_ctx *ctx, int32_t n_pics, const svt_yuv_planes *pics, int32_t pad_x, int32_t pad_y);

/* ------------------------------------------------------------------------------------------------------------------
 * Coefficient rate estimation -- SURVEY 8(f) row 4 (with T3, svt_hip_tq_batch_dist_device, this completes what
 * perform_dist_rate_calc computes per transform block, Codec/EbEncDecProcess.c:700-745).
 *
 * Replaces coeff_rate_estimate (Codec/EbRateDistortionCost.c:55-172, libvpx's cost_coeffs) with
 * use_fast_coef_costing = 0, the only way the reference calls it (:736-745): the bits (scaled by 2^9, vp9_cost.h) of
 * the tokens of one quantised transform block under the frame's coefficient probabilities.
 * Tables are the reference's own, handed over by the caller the way the scan tables are for TQ:
 *   token_costs   = x->token_costs (VPX/vp9_block.h:140, filled by fill_token_costs, VPX/vp9_rd.c:93-107)
 *   value_cost    = eb_vp9_dct_cat_lt_10_value_cost[v], v = -66 .. 66, stored at index v + 66 (VPX/vp9_tokenize.c:52)
 *   cat6_low_cost = eb_vp9_cat6_low_cost[256], cat6_high_cost = eb_vp9_cat6_high_cost[64] (VPX/vp9_tokenize.c:82, 97)
 * and per (tx_size, tx_type) the scan order's `scan` (n entries) followed by its `neighbors` (2 (n + 1) entries)
 * (VPX/vp9_scan.c), concatenated in one int16 array; a block names its table by element offset (scan_off). */
typedef struct svt_rate_tables {
    uint32_t token_costs[4][2][2][6][2][6][12]; /* [tx_size][plane_type][is_inter][band][prev token == 0][ctx][token] */
    int32_t  value_cost[133];
    uint16_t cat6_low_cost[256];
    uint16_t cat6_high_cost[64];
    uint16_t pad_[2];
} svt_rate_tables;                            /* 56472 bytes */

typedef struct svt_rate_block {
    uint32_t coeff_off;  /* element offset of the block's n*n quantised coefficients (raster) in d_qcoeff; multiple of 8
                            (the TQ batch's coeff_off) */
    uint32_t scan_off;   /* element offset of {scan[n], neighbors[2 (n + 1)]} of the block's scan order */
    uint16_t eob;
    uint8_t  tx_size;    /* SVT_TX_* */
    uint8_t  plane_type; /* 0 luma, 1 chroma (get_plane_type) */
    uint8_t  is_inter;   /* is_inter_block(mi) */
    uint8_t  ctx;        /* combine_entropy_contexts(left, above): 0..2 (EbEncDecProcess.c:734) */
    uint8_t  pad_[2];
} svt_rate_block;        /* 16 bytes */

/* d_bits[b] = coeff_rate_estimate(...) of block b.  Device pointers; blocks in any order.
 * d_scan must be 4-byte aligned and hold at least SVT_RATE_SCAN_MIN_ENTRIES int16 entries (the size of the four 4x4 and
 * four 8x8 tables): every workgroup stages that prefix in LDS before it looks at a block (with the canonical layout --
 * the 4x4 tables of tx_type 0..3 first, then the 8x8 ones -- those blocks read their scan from LDS; any other layout
 * is still correct, the entries are then read in place).  The host-pointer form pads a shorter array itself. */
#define SVT_RATE_SCAN_MIN_ENTRIES 976
int32_t svt_hip_coeff_rate_batch_device(svt_hip_ctx *ctx, const int16_t *d_qcoeff, const svt_rate_block *d_blocks,
                                        int32_t n_blocks, const svt_rate_tables *d_tables, const int16_t *d_scan,
                                        int32_t *d_bits);
/* Host-pointer convenience form. */
int32_t svt_hip_coeff_rate_batch(svt_hip_ctx *ctx, const int16_t *qcoeff, size_t coeff_count, const svt_rate_block *blocks,
                                 int32_t n_blocks, const svt_rate_tables *tables, const int16_t *scan, size_t scan_count,
                                 int32_t *bits);

/* ------------------------------------------------------------------------------------------------------------------
 * Picture-level EncDec: everything the encode pass does with mode decision's output, whole pictures at a time, device resident.
 *
 * Replaces, per batch of mutually independent pictures (e.g. the pictures of one temporal layer of a mini-GOP), the data path of
 * eb_vp9_enc_dec_kernel behind mode decision (Codec/EbEncDecProcess.c:5306): encode_pass_sb (:3627-4241) = inter prediction
 * (:3787-3802) -> perform_coding_loop per transform block (:3830, 3890, 3940) -> skip flags (:4069-4098) -> and, once per picture,
 * eb_vp9_build_mask_frame + eb_vp9_loop_filter_frame (:5651-5686) -> pad_ref_and_set_flags (:5694-5697).  The input is the mode-info
 * grid (what mode decision leaves in picture_control_set_ptr->mode_info_array), not block lists: the transform-block descriptors,
 * the skip flags and the LOOP_FILTER_MASKs are derived ON THE DEVICE (svt_tq_count / _emit, svt_tq_skip, svt_lf_mask kernels --
 * the same rules as svt_hip_tq_blocks_from_grid and svt_hip_lf_build_masks, one text: csrc/encdec_core.h), so no per-picture
 * descriptor data crosses PCIe and nothing waits for the host between the stages.
 * ------------------------------------------------------------------------------------------------------------------ */

/* Where the planes of one picture of a batch live relative to the batch's base pointers, and where its outputs go.  Offsets are
 * bytes from the batch's source / prediction base pointer (one 32-bit offset space each) and from the picture's own
 * reconstruction buffer; plane order Y, Cb, Cr. */
typedef struct svt_tq_pic_geom {
    uint32_t src_off[3], pred_off[3], recon_off[3];
    uint16_t src_stride[2], pred_stride[2], recon_stride[2]; /* luma, chroma */
    uint32_t coeff_base;    /* element offset of the picture's coefficient area (n_sb * SVT_SB_COEFFS elements) in the batch's arrays */
    int32_t  width, height; /* luma samples (multiples of 8) */
    uint8_t  recon_set;     /* which reconstruction base pointer of the batch (0..7): recon_off is relative to it */
    uint8_t  do_recon;
    uint8_t  pic;           /* index of the picture in its batch (0..63): goes into the position code */
    uint8_t  pad_[1];
} svt_tq_pic_geom;
/* Coefficient layout of the driver: per SB 64*64 luma + 2 x 32*32 chroma coefficients (Y at +0, Cb at +4096, Cr at +5120), every
 * block's N*N coefficients contiguous (raster inside the block) as in the reference's per-SB quantized_coeff_buffer
 * (Codec/EbEncDecProcess.c:4100-4108), but addressed by POSITION: inside a plane area the 4x4 units are in z-order, so the block at
 * sample (x, y) of its plane starts at zorder((x % S) / 4, (y % S) / 4) * 16 with S = 64 (luma) / 32 (chroma). */
#define SVT_SB_COEFFS 6144

/* host: the transform blocks of n_pics pictures (one geometry) from their mode-info grids, grouped by transform size and inside a
 * size ordered picture, SB (raster), 8x8 unit of the SB (raster), and per unit luma, Cb, Cr -- exactly the lists the driver builds
 * on the device.  The luma transform type of a block travels in svt_lf_mode_info.pad_[0] (0 = DCT_DCT).  pos[i] = picture-in-batch
 * << 24 | plane << 22 | (y / 4) << 11 | (x / 4) of block i, x / y in samples of its plane.  Returns the number of blocks or a negative error. */
int32_t svt_hip_tq_blocks_from_grid(int32_t n_pics, const svt_lf_mode_info *const *lf_mi, int32_t mi_stride, const svt_tq_pic_geom *geom,
                                    svt_tq_block *blocks, uint32_t *pos, int32_t capacity, int32_t size_count[4]);

/* host: the normative VP9 tables the transform stage needs (generated from the reference's own objects, tools/gen_vp9_tables.py):
 * the inverse scan orders eb_vp9_scan_orders[tx_size][tx_type].iscan (VPX/vp9_scan.c) concatenated, offsets16[tx_size * 4 +
 * tx_type] = element offset of a table; the quantiser steps eb_vp9_dc_quant / eb_vp9_ac_quant(q_index, 0, 8 bit)
 * (VPX/vp9_quant_common.c); eb_vp9_quantizer_to_qindex (VPX/vp9_quantize.c:329). */
const int16_t *svt_hip_vp9_iscan_tables(const uint32_t **offsets16, int32_t *entries);
int32_t svt_hip_vp9_qindex_from_qp(int32_t qp);
/* The q index a picture is coded at in fixed-QP mode: the sequence QP scaled by the picture's temporal layer -- QP_SCALING_MODE_0 of
 * eb_vp9_rate_control_kernel (Codec/EbRateControlProcess.c:4680-4722: eb_vp9_compute_qdelta with delta_rate_oq / _sq / _vmaf).  tune:
 * 0 SQ, 1 OQ, 2 VMAF; hierarchical_levels 3 or 4.  is_key: the sequence q index itself (key frames take the reference's adaptive
 * QP_SCALING_MODE_1, which is rate control proper and not part of this path). */
int32_t svt_hip_vp9_layer_qindex(int32_t qp, int32_t tune, int32_t hierarchical_levels, int32_t temporal_layer_index, int32_t is_key);
int32_t svt_hip_vp9_dc_step(int32_t q_index);
int32_t svt_hip_vp9_ac_step(int32_t q_index);
/* out[0] luma, out[1] chroma tables of a q index with zero deltas (the sequence-level eb_vp9_init_quantizer) */
int32_t svt_hip_quant_tables_for_qindex(int32_t q_index, svt_quant_tables out[2]);

/* Which stages the reference runs behind mode decision for a picture: the flags eb_vp9_signal_derivation_enc_dec_kernel_{sq,oq,
 * vmaf} derive (limit_intra, Codec/EbEncDecProcess.c:4989-4997 / 5127-5136 / 5257-5263; allow_enc_dec_mismatch, :4954-4959 /
 * 5084-5089 / 5201) and what encode_pass_sb (:3653-3657) and the picture's last SB (:5633-5697) do with them. */
typedef struct svt_encdec_flags_config {
    int32_t enc_mode, tune;
    int32_t temporal_layer_index;
    int32_t is_used_as_reference;
    int32_t recon_file;            /* static_config.recon_file */
    int32_t loop_filter;           /* static_config.loop_filter */
} svt_encdec_flags_config;
typedef struct svt_encdec_flags {
    int32_t limit_intra, allow_enc_dec_mismatch;
    int32_t do_recon;              /* inter SBs are reconstructed (inverse transform + add) */
    int32_t apply_loop_filter;     /* lf_application_enable_flag */
    int32_t pad_reference;         /* pad_ref_and_set_flags runs */
} svt_encdec_flags;
int32_t svt_hip_encdec_flags_derive(const svt_encdec_flags_config *cfg, svt_encdec_flags *flags);

/* One picture of a batch.  Every pointer is a device pointer.  The grids are [mi_rows][mi_stride] records per 8x8 unit. */
typedef struct svt_encdec_picture {
    const svt_mc_mode_info *d_mc_mi;   /* inter part of the mode info (prediction) */
    svt_lf_mode_info       *d_lf_mi;   /* sb_type, tx_size, is_inter, filter_level (+ luma tx_type in pad_[0]); `skip` is WRITTEN */
    svt_yuv_planes          src;       /* source picture, sample (0,0) pointers */
    svt_yuv_planes          ref[2];    /* padded reference pictures (as svt_mc_picture.ref) */
    svt_yuv_planes          pred;      /* prediction picture (written by the inter prediction, read by the transform stage) */
    svt_yuv_planes          recon;     /* the picture's reconstruction = reference buffer (sample (0,0) pointers into padded planes) */
    int16_t                *d_qcoeff, *d_dqcoeff; /* n_sb * SVT_SB_COEFFS each, 16-byte aligned, position-addressed (above).  d_dqcoeff may be
                                                      NULL (for every picture of a call or for none): the encode pass consumes the dequantised
                                                      coefficients in the lane that produced them (inverse transform) and nothing downstream
                                                      reads them -- they then never travel to memory (3 bytes per sample less traffic) */
    uint16_t               *d_eob_map; /* eob of every transform block at its 4x4 unit: [Y: (H/4) x (W/4)] [Cb: (H/8) x (W/8)] [Cr] */
    svt_lf_mask            *d_lfm;     /* [sb_rows][sb_cols] masks (written; read by the loop filter) */
    uint8_t                *d_nz;      /* scratch, mi_rows * mi_stride bytes */
    int32_t                 use_subpel; /* svt_mc_picture.use_subpel */
    int32_t                 no_pad;    /* 1: this picture is not padded although the batch's flags say pad_reference (a picture that is not
                                          used as a reference riding in a batch of reference pictures) */
    int32_t                 has_intra; /* 1: the grid holds intra blocks (is_inter = 0; sizes / modes as for svt_hip_encdec_intra_device): they are
                                          coded by the intra pass behind the batch's transform stage, from their neighbours' reconstruction
                                          (needs the flags' do_recon).  0: every block is inter; an intra block would be left uncoded */
    int32_t                 pad_;
} svt_encdec_picture;

/* workspace of a batch: descriptor lists, per-list eob, counters (device memory owned by the object); sized for max_pics pictures
 * of width x height.  Calls that share a workspace must be issued on the same context (stream order protects it). */
typedef struct svt_encdec_work svt_encdec_work;
int32_t svt_hip_encdec_work_create(svt_hip_ctx *ctx, int32_t max_pics, int32_t width, int32_t height, svt_encdec_work **work);
void    svt_hip_encdec_work_destroy(svt_hip_ctx *ctx, svt_encdec_work *work);

/* n_pics (<= 32, <= the workspace's max_pics) pictures of one geometry and one set of flags (the pictures of a temporal layer, or
 * of several layers of different mini-GOPs that share their flags; the reconstruction buffers of a batch must fall into at most 8
 * regions of 4 GB -- e.g. be carved out of a few slabs):
 * inter prediction -> transform blocks from the grids -> residual / transform / quantisation (/ reconstruction when
 * flags->do_recon) with the tables of q_index -> skip flags + eob map -> (flags->apply_loop_filter: masks + deblocking with
 * filter_level's thresholds) -> (flags->pad_reference: border of pad_x / pad_y samples).  Source and prediction planes of the
 * batch must each lie within 4 GB of the lowest of them (32-bit block offsets).  Asynchronous on the context's stream.
 * A malformed grid (block crossing the picture edge, transform larger than its block ...) is reported by
 * svt_hip_encdec_work_status after the work has completed; its blocks are left out. */
int32_t svt_hip_encdec_batch_device(svt_hip_ctx *ctx, svt_encdec_work *work, int32_t n_pics, const svt_encdec_picture *pics, int32_t width,
                                    int32_t height, int32_t mi_stride, int32_t q_index, const svt_encdec_flags *flags,
                                    const svt_lf_thresh *thr, int32_t pad_x, int32_t pad_y);
/* The encode pass of an INTRA picture (key frames, intra-refresh pictures): reference samples + the ten VP9 intra predictors +
 * transform / quantisation / reconstruction of every block in coding-dependency order (encode_pass_sb with intra blocks,
 * Codec/EbEncDecProcess.c:3680-4160: generate_intra_reference_samples :1128, intra_prediction Codec/EbIntraPrediction.c:16 ->
 * VPX/vp9_reconintra.c:249-408 / VPX/intrapred.c, perform_coding_loop :365), then the tail of svt_hip_encdec_batch_device: skip flags,
 * loop-filter masks, deblocking, border.  The grid (pic->d_lf_mi) describes intra blocks of 8x8, 16x16 or 32x32 (sb_type 3 / 6 / 9)
 * with the transform of their own size, is_inter = 0, pad_[1] = luma mode, pad_[2] = chroma mode (PREDICTION_MODE: 0 DC, 1 V, 2 H,
 * 3 D45, 4 D135, 5 D117, 6 D153, 7 D207, 8 D63, 9 TM); the luma transform type follows the mode
 * (eb_vp9_intra_mode_to_tx_type_lookup).  sb_type 0 = an 8x8 unit of four 4x4 luma blocks (+ one 4x4 chroma block per plane): tx_size 0,
 * the luma modes of blocks 0..3 in the nibbles of pad_[1] (blocks 0, 1) and pad_[0] (blocks 2, 3), pad_[2] the chroma mode.  The 64x64
 * block and rectangular blocks are outside this entry (reported as malformed through svt_hip_encdec_work_status).  pic->ref / d_mc_mi are ignored; pic->pred may be all NULL (the prediction is then not stored).
 * flags->do_recon must be set.  Planes and strides 4-byte aligned.  Asynchronous on the context's stream. */
int32_t svt_hip_encdec_intra_device(svt_hip_ctx *ctx, svt_encdec_work *work, const svt_encdec_picture *pic, int32_t width, int32_t height,
                                    int32_t mi_stride, int32_t q_index, const svt_encdec_flags *flags, const svt_lf_thresh *thr, int32_t pad_x,
                                    int32_t pad_y);
/* Stand-in decision for an intra picture (NOT the reference's): 16x16 blocks with DC prediction, 8x8 where a 16x16 block would
 * cross the picture edge. */
int32_t svt_hip_md_intra_default_device(svt_hip_ctx *ctx, int32_t width, int32_t height, int32_t filter_level, svt_lf_mode_info *d_lf_mi, int32_t mi_stride);
/* Profiling aid: `hook` is called on the enqueueing thread at every stage boundary of svt_hip_encdec_batch_device (before the
 * stage named is enqueued; SVT_ENCDEC_STAGE_END after the last), so that a host can record events of its own on the context's stream
 * and attribute the chain's time to its stages.  NULL removes it. */
#define SVT_ENCDEC_STAGE_MC 0
#define SVT_ENCDEC_STAGE_LISTS 1
#define SVT_ENCDEC_STAGE_TQ 2
#define SVT_ENCDEC_STAGE_SKIP 3
#define SVT_ENCDEC_STAGE_LF 4
#define SVT_ENCDEC_STAGE_PAD 5
#define SVT_ENCDEC_STAGE_END 6
typedef void (*svt_encdec_stage_hook)(void *user, int32_t stage);
void svt_hip_encdec_work_set_stage_hook(svt_encdec_work *work, svt_encdec_stage_hook hook, void *user);
/* synchronises the context; 0 = every grid the workspace has seen was well-formed, else SVT_HIP_ERR_BAD_PARAMETER (and the flag is
 * cleared).  counts[8] (optional) = offset and number of blocks per transform size of the most recent batch. */
int32_t svt_hip_encdec_work_status(svt_hip_ctx *ctx, svt_encdec_work *work, int32_t counts[8]);
/* most recent batch: copies the descriptor list, the position codes and the per-list eob to the host (capacity in blocks);
 * returns the number of blocks.  For tests and for hosts that walk the coefficients in list order. */
int32_t svt_hip_encdec_work_download(svt_hip_ctx *ctx, svt_encdec_work *work, svt_tq_block *blocks, uint32_t *pos, uint16_t *eob, int32_t capacity);

/* Stand-in for mode decision (NOT the reference's: mode decision is host control logic outside this path).  Turns the ME results
 * of n_pics pictures into well-formed mode-info grids so that the stages behind mode decision can run without a host-supplied
 * decision: bottom-up merge over the PU tree (16x16 -> 32x32 -> 64x64: a parent replaces its four children when
 * distortion(parent) <= sum distortion(children) + 3 lambda), blocks split at the picture edge down to 8x8, every block inter with
 * the direction / vectors of its own PU's best ME candidate and the transform of its own size (32x32 for 64x64).  d_results[i],
 * d_mc_mi[i], d_lf_mi[i] device arrays of picture i.  svt_hip_md_default_picture is the host form (same text). */
int32_t svt_hip_md_default_batch_device(svt_hip_ctx *ctx, int32_t n_pics, const svt_me_pu_result *const *d_results, int32_t width, int32_t height,
                                        uint32_t lambda, int32_t filter_level, svt_mc_mode_info *const *d_mc_mi, svt_lf_mode_info *const *d_lf_mi,
                                        int32_t mi_stride);
int32_t svt_hip_md_default_picture(const svt_me_pu_result *results, int32_t pic_width, int32_t pic_height, uint32_t lambda, int32_t filter_level,
                                   svt_mc_mode_info *mc_mi, svt_lf_mode_info *lf_mi, int32_t mi_stride);

/* device form of svt_hip_lf_build_masks (same text): masks of n_pics pictures from device-resident grids */
int32_t svt_hip_lf_build_masks_device(svt_hip_ctx *ctx, int32_t n_pics, const svt_lf_mode_info *const *d_lf_mi, int32_t mi_stride, int32_t mi_rows,
                                      int32_t mi_cols, svt_lf_mask *const *d_lfm);

/* Makes everything enqueued LATER on `ctx` wait for marker `marker` of `other` (both on the same device or not): the
 * stream-to-stream dependency a pipeline of contexts needs (ME one mini-GOP ahead of EncDec) without a host wait. */
int32_t svt_hip_ctx_wait_marker(svt_hip_ctx *ctx, svt_hip_ctx *other, uint64_t marker);
/* asynchronous device-to-host copy into PINNED host memory (svt_hip_host_alloc), ordered on the context's stream; completion is
 * observed through a marker recorded after it */
int32_t svt_hip_host_alloc(svt_hip_ctx *ctx, size_t bytes, void **ptr);
void    svt_hip_host_free(svt_hip_ctx *ctx, void *ptr);
int32_t svt_hip_mem_download_2d_async(svt_hip_ctx *ctx, void *pinned_dst, size_t dst_stride, const void *d_src, size_t src_stride, size_t width_bytes,
                                      size_t rows);
/* device-to-device rectangle copy on the context's stream */
int32_t svt_hip_mem_copy_2d_device(svt_hip_ctx *ctx, void *d_dst, size_t dst_stride, const void *d_src, size_t src_stride, size_t width_bytes,
                                   size_t rows);

/* ------------------------------------------------------------------------------------------------ */
/* Multi-GPU: GOP sharding and the inter-segment reference hand-off (scope row e)                      */
/* ------------------------------------------------------------------------------------------------ */
/* Closed GOPs are independent units of the reference (every intra refresh is a key frame, Codec/EbPictureDecisionProcess.c:952,
 * 1596-1603): GOP g is encoded by device g % n_devices and the data path needs no exchange.  svt_hip_gop_assign writes the
 * GOPs of device `index` (ascending) into gops[] (capacity max) and returns how many there are; svt_hip_gop_owner is g % n. */
int32_t svt_hip_gop_owner(int64_t gop, int32_t n_devices);
int32_t svt_hip_gop_assign(int64_t n_gops, int32_t n_devices, int32_t index, int64_t *gops, int32_t max);
/* Split-GOP (low-latency) mode: consecutive mini-GOPs of ONE GOP go to consecutive devices; mini-GOP m is encoded by device
 * m % n_devices and needs the reconstructed, deblocked, padded base-layer picture of mini-GOP m - 1 from device (m - 1) %
 * n_devices first -- the one exchange step of the path (a point-to-point copy over one xGMI link, ~13.9 MB at 4K).
 * Returns the device that produces the reference mini-GOP m needs, or -1 for m = 0 (it starts from the key frame). */
int32_t svt_hip_minigop_reference_source(int64_t minigop, int32_t n_devices);

/* How the reference cuts the pictures waiting in its pre-assignment buffer when they are released short of a full mini-GOP (end of
 * stream, or an intra refresh arrived) into the units it assigns prediction structures to: eb_vp9_generate_picture_window_split +
 * eb_vp9_handle_incomplete_picture_window_map as the picture-decision kernel drives them (Codec/EbPictureDecisionProcess.c:387-476,
 * 1662-1680; mini-GOP table Codec/EbUtility.c:167-185).  A full group (n_pictures == 1 << hierarchical_levels) is one part.
 * cut_by_intra != 0: the group was released by an intra refresh and n_pictures INCLUDES that intra picture as its last element, as
 * the reference's pre_assignment_buffer_count does (:1641-1646).  A part whose length equals the period of its own levels (8
 * pictures at 3 levels) keeps the random-access hierarchy; any other part is coded with the low-delay P structure (:1711-1727), and
 * so is the LAST part of a group cut by an intra refresh -- the one that ends with the intra picture (mini_gop_idr_count is set for
 * the last part only, :419-424, 463-472): earlier whole-period parts of such a group stay random access.
 * Returns the number of parts (<= 4) or a negative error. */
typedef struct svt_minigop_part {
    int32_t start, length;          /* pictures [start, start + length) of the group */
    int32_t hierarchical_levels;    /* of the part (mini_gop_hierarchical_levels) */
    int32_t random_access;          /* 1: hierarchical B pictures; 0: low-delay P */
} svt_minigop_part;
int32_t svt_hip_minigop_split(int32_t n_pictures, int32_t hierarchical_levels, int32_t cut_by_intra, svt_minigop_part parts[4]);

/* A set of contexts, one per device of a node, for a host that drives several GPUs from one process (one host thread per
 * device is the intended use: a context is not shared between threads).  Devices are HIP ordinals. */
typedef struct svt_hip_device_set svt_hip_device_set;
int32_t      svt_hip_device_set_create(svt_hip_device_set **set, const int32_t *device_ordinals, int32_t n_devices);
int32_t      svt_hip_device_set_size(const svt_hip_device_set *set);
svt_hip_ctx *svt_hip_device_set_ctx(svt_hip_device_set *set, int32_t index);
void         svt_hip_device_set_destroy(svt_hip_device_set *set);
/* The hand-off itself: `bytes` of device memory (a padded picture buffer) from d_src on src's device to d_dst on dst's
 * device, ordered after everything enqueued on src's stream and before everything enqueued later on dst's stream; src's
 * stream also waits for the copy, so the producer may overwrite d_src afterwards.  Peer access is enabled on first use;
 * the copy travels device to device (xGMI) when the devices are peers.  src == dst degenerates to a device-local copy. */
int32_t svt_hip_ref_handoff_device(svt_hip_ctx *src, const void *d_src, svt_hip_ctx *dst, void *d_dst, size_t bytes);

/* ------------------------------------------------------------------------------------------------ */
/* IVF container (host only): what the reference's sample application wraps the encoder's packets in   */
/* ------------------------------------------------------------------------------------------------ */
#define SVT_IVF_STREAM_HEADER_BYTES 32
/* replaces write_ivf_stream_header (App/EbAppProcessCmd.c:515-540); frame_rate_q16 is used when numerator or
 * denominator is 0, exactly as there */
int32_t svt_ivf_stream_header(uint8_t out[SVT_IVF_STREAM_HEADER_BYTES], uint32_t width, uint32_t height, uint32_t frame_rate_q16,
                              uint32_t rate_numerator, uint32_t rate_denominator);
/* replaces the stream-writing branch of process_output_stream_buffer (App/EbAppProcessCmd.c:617-650): one encoder packet
 * -> one IVF frame, or five when the packet is flagged EB_BUFFERFLAG_SHOW_EXT (the coded frame, then the four trailing
 * one-byte show-existing-frame headers with pts - 2, - 1, + 0, + 1).  Returns bytes written or a negative error. */
int64_t svt_ivf_packetize(const uint8_t *packet, uint32_t len, uint64_t pts, int32_t show_ext, uint8_t *out, size_t capacity);

#ifdef __cplusplus
}
#endif
#endif /* SVTVP9_HIP_H */
