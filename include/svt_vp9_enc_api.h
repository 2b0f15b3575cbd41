/*
 * svt_vp9_enc_api.h -- the public C ABI of the SVT-VP9 encoder library as exported by libSvtVp9Enc.so of this repository
 * (scope row b-1).
 *
 * An application written against the reference's Source/API/EbSvtVp9Enc.h links this library unchanged: the eleven entry
 * points, the four structures (same field order, sizes and offsets -- pinned in tests/test_enc_shim.py against the
 * reference header compiled in the build container), the error codes and the buffer flags are the reference's
 * (API/EbSvtVp9Enc.h:50-120 structures, :124-355 configuration, :365-439 functions).  This header is this repository's own
 * statement of that ABI; the reference's header can be used in its place.
 *
 * What the library does behind the ABI is the hot path of this repository, not an encoder: pictures are copied to the GPU
 * (the caller may reuse its buffers as soon as eb_vp9_svt_enc_send_picture returns, Codec/EbEncHandle.c:2743-2796), grouped
 * into the reference's mini-GOPs and run through picture analysis, motion estimation and -- behind a mode decision that is
 * either the host's (svt_vp9_shim_set_mode_decision) or a built-in stand-in -- inter prediction, transform / quantisation /
 * reconstruction, in-loop deblocking and reference padding, every picture predicting from the reconstructed reference pictures
 * before it (the closed loop of Codec/EbEncDecProcess.c).  Entropy coding is out of scope, so every picture is reported by a
 * packet of ZERO bytes (n_filled_len = 0, p_buffer = NULL) carrying its pts and, on the last one, EB_BUFFERFLAG_EOS -- no
 * bitstream is produced and none is pretended.  eb_vp9_svt_get_recon delivers the reconstructed pictures as the reference
 * does (Codec/EbEncHandle.c:2837-2865, recon_output Codec/EbEncDecProcess.c:4693-4820): with recon_file != 0, one picture per
 * call in coding order, pts = picture number, W x H luma then the two chroma planes, EB_BUFFERFLAG_EOS on the last;
 * EB_NoErrorEmptyQueue while none is ready; EB_ErrorMax when recon_file is 0.
 */
#ifndef SVT_VP9_ENC_API_H
#define SVT_VP9_ENC_API_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define EB_BUFFERFLAG_EOS 0x00000001u      /* last packet of the stream */
#define EB_BUFFERFLAG_SHOW_EXT 0x00000002u /* packet ends with show-existing-frame headers (never set by this library) */

typedef uint8_t EbBool;

typedef struct EbComponentType {
    uint32_t n_size;                /* sizeof(EbComponentType), written by the library */
    void    *p_component_private;   /* library state */
    void    *p_application_private; /* the p_app_data of eb_vp9_svt_init_handle */
} EbComponentType;

/* picture handed to eb_vp9_svt_enc_send_picture through EbBufferHeaderType.p_buffer: 8-bit 4:2:0 planes */
typedef struct EbSvtEncInput {
    uint8_t *luma, *cb, *cr;
    uint8_t *luma_ext, *cb_ext, *cr_ext; /* 10-bit extension planes: unused (8-bit only) */
    uint32_t y_stride, cr_stride, cb_stride;
} EbSvtEncInput;

typedef struct EbBufferHeaderType {
    uint32_t size;
    uint8_t *p_buffer;
    uint32_t n_filled_len, n_alloc_len;
    void    *p_app_private, *wrapper_ptr;
    uint32_t n_tick_count;
    int64_t  dts, pts;
    uint32_t qp, pic_type, flags;
} EbBufferHeaderType;

typedef enum EbErrorType {
    EB_ErrorNone                  = 0,
    EB_ErrorInsufficientResources = (int32_t)0x80001000,
    EB_ErrorUndefined             = (int32_t)0x80001001,
    EB_ErrorInvalidComponent      = (int32_t)0x80001004,
    EB_ErrorBadParameter          = (int32_t)0x80001005,
    EB_NoErrorEmptyQueue          = (int32_t)0x80002033,
    EB_ErrorMax                   = 0x7FFFFFFF
} EbErrorType;

typedef struct EbSvtVp9EncConfiguration {
    uint8_t  enc_mode;               /* 0..12 (limits by resolution, see eb_vp9_svt_enc_set_parameter) */
    uint8_t  tune;                   /* 0 SQ, 1 OQ, 2 VMAF */
    int32_t  intra_period;           /* -2 automatic, -1 none, 0..255 */
    uint8_t  pred_structure;         /* 2 = random access, the only one accepted */
    uint32_t base_layer_switch_mode;
    uint32_t source_width, source_height;
    uint32_t frame_rate;             /* Q16 when > 1000 */
    uint32_t frame_rate_numerator, frame_rate_denominator;
    uint32_t encoder_bit_depth;      /* 8 */
    uint32_t partition_depth;
    uint32_t qp;                     /* 0..63 */
    EbBool   use_qp_file;
    uint32_t enable_qp_scaling_flag;
    EbBool   loop_filter;
    EbBool   use_default_me_hme;
    EbBool   enable_hme_flag;
    uint32_t search_area_width, search_area_height;
    uint32_t rate_control_mode;      /* 0 CQP, 1 VBR, 2 CBR */
    uint32_t target_bit_rate, max_qp_allowed, min_qp_allowed;
    uint32_t profile, level, asm_type;
    uint32_t channel_id, active_channel_count, speed_control_flag;
    int32_t  injector_frame_rate;
    uint32_t logical_processors;
    int32_t  target_socket;
    uint32_t recon_file;
    uint32_t input_picture_stride;
    uint32_t vbv_max_rate, vbv_buf_size;
    uint64_t frames_to_be_encoded;
} EbSvtVp9EncConfiguration;

/* allocates the handle and loads the library defaults into *config_ptr (Codec/EbEncHandle.c:1762-1852) */
EbErrorType eb_vp9_svt_init_handle(EbComponentType **p_handle, void *p_app_data, EbSvtVp9EncConfiguration *config_ptr);
/* copies and checks the configuration (the reference's verify_settings rules, :2203-2557) */
EbErrorType eb_vp9_svt_enc_set_parameter(EbComponentType *svt_enc_component, EbSvtVp9EncConfiguration *p_component_parameter_structure);
/* acquires the GPU and the picture buffers; EB_ErrorInsufficientResources without a usable device */
EbErrorType eb_vp9_init_encoder(EbComponentType *svt_enc_component);
EbErrorType eb_vp9_svt_enc_stream_header(EbComponentType *svt_enc_component, EbBufferHeaderType **output_stream_ptr); /* no-op, as in the reference */
EbErrorType eb_vp9_svt_enc_eos_nal(EbComponentType *svt_enc_component, EbBufferHeaderType **output_stream_ptr);       /* no-op, as in the reference */
EbErrorType eb_vp9_svt_enc_send_picture(EbComponentType *svt_enc_component, EbBufferHeaderType *p_buffer);
EbErrorType eb_vp9_svt_get_packet(EbComponentType *svt_enc_component, EbBufferHeaderType **p_buffer, uint8_t pic_send_done);
void        eb_vp9_svt_release_out_buffer(EbBufferHeaderType **p_buffer);
EbErrorType eb_vp9_svt_get_recon(EbComponentType *svt_enc_component, EbBufferHeaderType *p_buffer);
EbErrorType eb_vp9_deinit_encoder(EbComponentType *svt_enc_component);
EbErrorType eb_vp9_deinit_handle(EbComponentType *svt_enc_component);

/* ---- extension of this repository (not part of the reference ABI): what the hot path computed for a picture ---- */
typedef struct svt_vp9_shim_picture_info {
    uint64_t picture_number;       /* display order, 0-based */
    int32_t  is_intra;             /* no motion estimation ran */
    int32_t  temporal_layer_index, hierarchical_levels, num_ref_lists;
    int64_t  ref_picture_number[2]; /* display-order numbers of the list 0 / list 1 reference pictures (-1: none) */
    uint32_t n_sb;
    /* the stages behind mode decision (svt_encdec_flags of svtvp9_hip.h, as the reference derives them for this picture) */
    int32_t  is_used_as_reference, do_recon, apply_loop_filter, pad_reference;
    int32_t  q_index, filter_level;
    int32_t  decision_source;      /* 0 built-in stand-in, 1 the host's callback */
    int32_t  intra_recon_is_source; /* always 0 since round 4: intra pictures go through the intra encode pass on the GPU
                                      (svt_hip_encdec_intra_device); the field keeps the struct layout of round 3 */
    int32_t  device_ordinal;       /* the GPU that coded the picture's GOP */
} svt_vp9_shim_picture_info;
/* copies the ME results of a picture whose mini-GOP has been processed (and not yet overwritten: the library keeps the last two
 * mini-GOPs + 2 pictures of a GOP's device) into out (n_sb * 85 records of 40 bytes, svt_me_pu_result of svtvp9_hip.h); waits for
 * the GPU work of that picture.  EB_NoErrorEmptyQueue while the picture still waits in an incomplete mini-GOP (or is no longer
 * kept). */
EbErrorType svt_vp9_shim_get_me_results(EbComponentType *svt_enc_component, uint64_t picture_number, svt_vp9_shim_picture_info *info,
                                        void *out, uint64_t out_bytes);
/* the per-SB side outputs of the same picture: stats = n_sb records of svt_me_sb_stats (8 bytes), histograms = 257 uint32 (ME
 * distortion histogram, OIS histogram, number of complete SBs), mean / variance = n_sb * 85 picture-analysis values; any pointer
 * may be NULL */
EbErrorType svt_vp9_shim_get_sb_stats(EbComponentType *svt_enc_component, uint64_t picture_number, void *stats, uint64_t stats_bytes,
                                      uint32_t *histograms, uint8_t *mean, uint16_t *variance);
/* how many batched ME launches the library has issued, how many pictures it has accepted */
EbErrorType svt_vp9_shim_get_counters(EbComponentType *svt_enc_component, uint64_t *me_launches, uint64_t *pictures_sent);

/* Mode decision is the host's (control logic outside the hot path).  A host that has one registers it here, before
 * eb_vp9_init_encoder: the library calls it once per picture, in coding order, on the thread that calls
 * eb_vp9_svt_enc_send_picture, after the picture's motion estimation has completed (the call waits for it: a host decision
 * serialises the GPU pipeline at this point, which is what a host-side decision costs), with the picture's ME results
 * (n_sb * 85 records of svt_me_pu_result).  It fills the two mode-info grids of svtvp9_hip.h -- mc_mode_info: (height / 8) rows of
 * mi_stride records of svt_mc_mode_info (12 bytes); lf_mode_info: the same grid of svt_lf_mode_info (8 bytes; `skip` is the
 * library's to write) -- and returns 0; a non-zero return makes the library use its stand-in for that picture.  Without a
 * callback the stand-in decides every picture (svt_hip_md_default_batch_device: a deterministic partition from the ME results,
 * NOT the reference's mode decision).  For an intra picture (info->is_intra) me_results is NULL and only lf_mode_info is read: intra
 * blocks of 8x8 / 16x16 / 32x32 (sb_type 3 / 6 / 9, tx_size 1 / 2 / 3, is_inter 0) with pad_[1] = luma mode and pad_[2] = chroma mode;
 * units of four 4x4 blocks: sb_type 0, tx_size 0, their luma modes in the nibbles of pad_[1] (blocks 0, 1) and pad_[0] (blocks 2, 3)
 * (0 DC, 1 V, 2 H, 3 D45, 4 D135, 5 D117, 6 D153, 7 D207, 8 D63, 9 TM), svt_hip_encdec_intra_device of svtvp9_hip.h; the stand-in
 * for an intra picture is 16x16 blocks with DC prediction.  The grids of an INTER picture may hold intra blocks as well (is_inter 0 + modes in
 * lf_mode_info, ref_list[0] = -1 in mc_mode_info) as long as the picture is reconstructed (info->do_recon; the reference's limit_intra
 * forbids intra blocks in the others): they are coded from their neighbours' reconstruction behind the inter blocks. */
typedef int32_t (*svt_vp9_shim_md_callback)(void *user, const svt_vp9_shim_picture_info *info, const void *me_results, void *mc_mode_info,
                                            void *lf_mode_info, int32_t mi_stride);
EbErrorType svt_vp9_shim_set_mode_decision(EbComponentType *svt_enc_component, svt_vp9_shim_md_callback callback, void *user);
/* what the stages behind mode decision left for a coded picture (kept as long as its ME results): the two grids (lf grid with the
 * skip flags the transform stage produced), the quantised coefficients (n_sb * 6144 int16, position-addressed: SVT_SB_COEFFS
 * layout of svtvp9_hip.h), the eob of every transform block at its 4x4 unit (Y (H/4 x W/4), Cb, Cr; uint16).  Any pointer may be
 * NULL.  EB_NoErrorEmptyQueue while the picture has not been coded (or is no longer kept). */
EbErrorType svt_vp9_shim_get_coded_picture(EbComponentType *svt_enc_component, uint64_t picture_number, svt_vp9_shim_picture_info *info,
                                           void *mc_mode_info, void *lf_mode_info, int16_t *qcoeff, uint16_t *eob_map);
/* the padded reference picture of a coded picture as later pictures predict from it: three planes, Y (W + 160) x (H + 160), then
 * Cb and Cr ((W / 2 + 80) x (H / 2 + 80)); bytes must be at least their sum */
EbErrorType svt_vp9_shim_get_reference_picture(EbComponentType *svt_enc_component, uint64_t picture_number, uint8_t *out, uint64_t bytes);

#ifdef __cplusplus
}
#endif
#endif
