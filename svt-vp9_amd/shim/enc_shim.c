/*
 * enc_shim.c -- libSvtVp9Enc.so: the reference's public encoder ABI (Source/API/EbSvtVp9Enc.h:365-439) in front of the GPU hot
 * path of this repository.  Plain C; the GPU is reached through the C ABI of libsvtvp9_hip.so only.
 *
 * What mirrors the reference, with the lines it follows:
 *   eb_vp9_svt_init_handle       handle malloc'd by the library, *config_ptr overwritten with the defaults
 *                                (Codec/EbEncHandle.c:1762-1852: eb_vp9_svt_enc_init_parameter; fields it does not touch stay untouched)
 *   eb_vp9_svt_enc_set_parameter copy + verify_settings' rules (:2052-2200, 2203-2557), hierarchical levels and automatic intra
 *                                period of set_param_based_on_input (:2166-2192)
 *   eb_vp9_svt_enc_send_picture  the picture is COPIED before the call returns (:2743-2796); NULL p_buffer / EOS flag ends the stream
 *   eb_vp9_svt_get_packet        non-blocking poll -> EB_NoErrorEmptyQueue when nothing is ready (:2880-2915); packets are the
 *                                library's until eb_vp9_svt_release_out_buffer (:1752-1757)
 *   eb_vp9_svt_get_recon         with recon_file: one reconstructed picture per call in coding order, pts = picture number, EOS on
 *                                the last (:2837-2865 <- recon_output, Codec/EbEncDecProcess.c:4693-4820); EB_ErrorMax without
 *   stream_header / eos_nal      no-ops returning EB_ErrorNone (:2953-2971)
 *
 * What the library does with the pictures (all of it enqueued asynchronously; send_picture returns after the host copy of the
 * picture into pinned staging):
 *   as each picture arrives      three planes to the device (one staging copy, one transfer, on an UPLOAD context), picture analysis
 *                                (padded / decimated luma planes, block mean / variance) on an INPUT context behind it -- streams of their
 *                                own, so that the next mini-GOPs' pictures cross PCIe while the current one computes; a picture slot (the
 *                                ring holds SVT_HIP_RING_GROUPS mini-GOPs) is overwritten behind the marker of its last reader on the main
 *                                context (device-side waits, svt_hip_ctx_wait_marker); the reconstruction copies of get_recon run on
 *                                an OUTPUT context the same way (SVT_HIP_SINGLE_STREAM=1: everything on the main stream)
 *   intra pictures               coded by the intra encode pass (svt_hip_encdec_intra_device: wavefront prediction + transform, then
 *                                deblocking and border); the decision callback is asked for them too, the stand-in is 16x16 / DC; on a KEY
 *                                context of their own, beside the previous GOP's tail and the new GOP's motion estimation
 *   when a mini-GOP is complete  (or cut short by an intra refresh / the end of the stream: cut as the reference cuts it,
 *                                svt_hip_minigop_split) the caller's thread plans the group (structure, packets queued in decode order) and
 *                                the device's FEEDER thread enqueues it while the caller goes on with the next pictures: ONE batched motion-estimation launch for all its pictures + the per-SB ME
 *                                statistics, then the stages behind mode decision in dependency order -- one batch per temporal
 *                                layer: mode decision (the host's callback, or the built-in stand-in) -> inter prediction from the
 *                                reconstructed, padded reference pictures -> transform / quantisation / reconstruction -> skip flags
 *                                -> deblocking -> reference padding (svt_hip_encdec_batch_device), with the per-picture stage flags
 *                                the reference derives (svt_hip_encdec_flags_derive)
 *   GOPs                         closed GOPs are independent (Codec/EbPictureDecisionProcess.c:952): with SVT_HIP_DEVICES=0,1,.. GOP g
 *                                is coded on device g mod N, each with its own context and picture ring; with
 *                                SVT_HIP_SPLIT_GOP=1 (latency mode) consecutive MINI-GOPs go to consecutive devices instead and the
 *                                padded base-layer reconstruction (+ its analysed planes) is handed to the next device, device to
 *                                device (svt_hip_ref_handoff_device): the one exchange step of the path
 * Every picture is answered by a zero-byte packet: entropy coding is outside the hot path.
 * Not reproduced (picture decision / rate control, control plane): the low-delay-P structure tables of the parts of a short group
 * (those pictures are a P chain); the q index of KEY frames -- inter pictures follow the reference's fixed-QP rule per temporal layer
 * (QP_SCALING_MODE_0, host/qp_scaling.c), key frames are coded at the sequence's q index where the reference applies its adaptive
 * QP_SCALING_MODE_1 to I slices (Codec/EbRateControlProcess.c:4680-4722: a noticeably lower q): pictures that predict from a key frame
 * see a coarser reference here than upstream.  Rate control's decision: out of scope.
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define FAILED(s) __atomic_load_n(&(s)->failed, __ATOMIC_RELAXED)
#define SET_FAILED(s) __atomic_store_n(&(s)->failed, 1, __ATOMIC_RELAXED)
#include <time.h>

#include "../../include/svt_vp9_enc_api.h"
#include "../../include/svtvp9_hip.h"
#include "../csrc/encdec_core.h" /* svt_tq_unit_is_origin: the block rules the device-side driver applies to a mode-info grid */

/* host/copy_pool.c of libsvtvp9_hip.so: rows copied by a few threads (the staging copy of the uploads uses it too) */
void svt_copy_rows_mt(uint8_t *dst, size_t dst_stride, const uint8_t *src, size_t src_stride, size_t width, size_t rows);

#define SHIM_MAX_MINIGOP 16
#define SHIM_MAX_DEV 8
#define SHIM_REF_PAD 80 /* border of a reference picture: 64 + 16 (Codec/EbEncHandle.c:968-971) */
#define SHIM_WAVE_MAX 8 /* pictures per EncDec batch (the deepest temporal layer of a 16-picture mini-GOP) */

typedef struct shim_packet {
    EbBufferHeaderType  hdr;
    struct shim_packet *next;
    int                 dev;
    svt_hip_ctx        *ctx;        /* the context the marker belongs to (NULL: the device's main context) */
    uint64_t            marker;     /* the GPU work behind this packet (svt_hip_ctx_marker_*) */
    int                 ready;      /* ctx / marker are valid: a packet is queued (in decode order) when its group is planned and completed by
                                       the device's feeder once the group's work has been enqueued (release / acquire) */
} shim_packet;

typedef struct shim_recon { /* one reconstructed picture on its way to eb_vp9_svt_get_recon: pinned host memory, filled asynchronously */
    struct shim_recon *next;
    uint8_t           *host;
    uint8_t           *d_tight;   /* device: the picture's three planes packed tight (Y | Cb | Cr), what ONE device-to-host transfer then carries */
    int                dev;
    uint64_t           marker;
    int64_t            pts;
    uint32_t           flags;
    int                ready;      /* as shim_packet.ready */
} shim_recon;

typedef struct shim_slot { /* one buffered picture, everything device resident */
    int64_t        number;  /* display order, -1 = empty */
    int64_t        pts;
    svt_pa_picture pa;
    uint8_t       *d_src;   /* Y (W x H) then Cb then Cr, tight: inside the device's source slab */
    uint8_t       *d_pred;  /* prediction picture, same layout: inside the prediction slab */
    uint8_t       *d_rec;   /* reconstruction = reference picture, three padded planes */
    void          *d_results, *d_mean, *d_var, *d_rcme, *d_stats, *d_hist;
    void          *d_mc_mi, *d_lf_mi, *d_eob_map, *d_lfm, *d_nz;
    int16_t       *d_qcoeff, *d_dqcoeff;
    int            processed;   /* its ME (or, for an intra picture, its analysis) has been enqueued */
    int            coded;       /* the stages behind mode decision have been enqueued */
    int            has_marker;
    uint64_t       release;      /* marker of the main context after which nothing enqueued so far reads this slot's buffers */
    int            has_release;
    uint64_t       out_marker;   /* ctx_out: the device-to-host copy of this slot's reconstruction */
    int            has_out;
    int            is_copy;     /* split-GOP mode: the base picture of the previous mini-GOP, handed over from the device that coded it
                                   (analysed planes + reference picture only) */
    uint64_t       marker;      /* completion of everything enqueued for this picture so far */
    svt_hip_ctx   *marker_ctx;  /* the context `marker` belongs to (NULL: the device's main context) */
    uint64_t       release2;    /* the same as `release` for the deep-layer context (ctx_deep) */
    int            has_release2;
    svt_vp9_shim_picture_info info;
} shim_slot;

/* one picture of a group whose motion estimation is about to be launched */
typedef struct shim_job {
    int64_t       number, ref0, ref1;
    int           layer, levels, n_lists, used_as_ref, wave;
    svt_me_params p;
} shim_job;

/* A planned group: everything the enqueuing side needs, fixed by the caller's thread (plan_group) -- the pictures in decode order with
 * their references, waves and ME parameters, the packets (and reconstruction records) already queued in output order, the input
 * stream's marker behind the group's last picture.  run_group enqueues it: on the caller's thread, or on the device's feeder. */
typedef struct shim_group {
    shim_job     jobs[SHIM_MAX_MINIGOP];
    shim_packet *pkt[SHIM_MAX_MINIGOP];
    shim_recon  *rec[SHIM_MAX_MINIGOP];
    int          n, n_waves, end_of_stream, dev;
    int          has_in;
    uint64_t     in_marker;
    int          has_key;               /* the group's waves follow the intra picture coded on the key context */
    uint64_t     key_marker;
    int64_t      first, last, oldest;   /* first / last picture of the group; the oldest picture its work reads (a reference) */
} shim_group;

struct shim_state;
typedef struct shim_dev {
    svt_hip_ctx     *ctx;
    /* input side: uploads and picture analysis run on a context (stream) of their own, so that the pictures of mini-GOP k + 1 cross
       PCIe while the stages of mini-GOP k compute; the two sides meet through markers (svt_hip_ctx_wait_marker, no host wait).
       SVT_HIP_SINGLE_STREAM=1 makes it the main context again (everything in one stream, as in round 3). */
    svt_hip_ctx     *ctx_in;
    svt_hip_ctx     *ctx_up;         /* the uploads themselves: a stream of nothing but host-to-device copies, back to back (the analysis kernels and
                                        their descriptor copies wait on ctx_in behind each picture's copy: ~60 us of stream latency per picture that the
                                        link would otherwise idle through).  SVT_HIP_SINGLE_STREAM=1: the main context */
    uint64_t         in_marker;      /* ctx_in: the latest picture's upload + analysis */
    int              has_in;
    svt_hip_ctx     *ctx_out;        /* output side: the reconstructions' device-to-host copies (recon_file), behind the main stream's markers */
    /* the two deepest temporal layers of a group (12 of a mini-GOP's 16 pictures) are coded on a context of their own, behind the
       group's shallower layers: nothing of the NEXT group depends on them (its base picture predicts from this group's base picture),
       so the next group's motion estimation and shallow layers -- waves of 1, 1 and 2 pictures, bound by the deblocking wavefront's
       latency -- run beside them instead of behind them.  Opt-in (SVT_HIP_DEEP_STREAM=1), see eb_vp9_init_encoder; default: the main context. */
    svt_hip_ctx     *ctx_deep;
    svt_hip_ctx     *ctx_me;         /* motion estimation + statistics of a group on a stream of their own (the next group's search runs beside this group's
                                        layers): the default without reconstructed output, SVT_HIP_ME_STREAM=0 / 1 overrides; else the main context */
    svt_encdec_work *work_deep;
    /* a key frame's intra encode pass (a dependency wavefront over the picture: ~6 ms at 4K on a fraction of the device) runs on a context
       of its own: the motion estimation of the GOP's first group -- which reads the key frame's analysed planes, not its reconstruction --
       and the tail of the previous GOP run beside it; the group's first wave waits for it (key_marker).  SVT_HIP_NO_KEY_STREAM=1, a
       decision callback or SVT_HIP_SINGLE_STREAM=1: the main context. */
    svt_hip_ctx     *ctx_key;
    svt_encdec_work *work_key;
    uint64_t         key_marker;     /* ctx_key: the latest intra picture (caller's thread; a planned group carries its copy) */
    int              has_key;
    int              ordinal;
    int              n_slots;
    shim_slot       *slot;
    int64_t          accepted;       /* pictures this device has taken: ring position */
    void            *d_src_slab, *d_pred_slab, *d_q_slab, *d_dq_slab;
    svt_encdec_work *work;
    shim_recon      *free_recon;     /* pinned buffers ready for re-use */
    /* The device's feeder: a thread that enqueues the planned groups of this device (motion estimation, statistics, the waves behind
       mode decision: ~2/3 of the stream operations of a picture) while the caller's thread goes on copying and uploading the next
       pictures -- the public API's picture rate is bound by the operations ONE thread can enqueue, and with several devices one thread
       cannot feed them all (the reference's pipeline has a thread per process for the same reason, Codec/EbEncHandle.c:1901-2012).
       One group per device at a time: the caller joins the previous one before it posts the next, and before anything else of its
       own that touches the device's main contexts or the group's slots.  SVT_HIP_NO_FEEDER=1: everything on the caller's thread. */
    struct shim_state *owner;
    pthread_t        feeder;
    int              has_feeder;
    pthread_mutex_t  mu;
    pthread_cond_t   cv;
    int              job_state;      /* 0 idle, 1 posted, 2 running (under mu) */
    int              stop;
    int              job_result;     /* EbErrorType of the last group (under mu) */
    int              outstanding;    /* caller's side: a group has been posted and not joined yet */
    shim_group       group;
} shim_dev;

typedef struct shim_state {
    EbSvtVp9EncConfiguration cfg;   /* the library's copy, with frame_rate / intra_period resolved (copy_api_from_app) */
    int         configured, initialised, eos;
    int         failed;      /* written by the caller's thread and by the feeders: accessed through FAILED() / SET_FAILED() (relaxed atomics) */
    int         levels, minigop;       /* hierarchical levels, 1 << levels */
    int         intra_period;          /* resolved */
    int         n_dev, cur_dev, split_gop;
    int         register_input;        /* SVT_HIP_REGISTER_INPUT=1 */
    int         use_feeder;            /* per-device feeder threads (off with a decision callback, in split-GOP mode, SVT_HIP_NO_FEEDER=1) */
    int         profile;               /* SVT_HIP_SHIM_PROFILE=1: host time per section, printed by eb_vp9_deinit_encoder */
    double      prof_s[8];
    shim_dev    dev[SHIM_MAX_DEV];
    int64_t     gop;                   /* index of the GOP being sent */
    int64_t     next_number;           /* display number of the next picture sent */
    int64_t     pending_first;         /* first picture of the mini-GOP being collected */
    int         pending;               /* pictures collected */
    int64_t     last_base;             /* display number of the latest base-layer / intra picture of the current GOP (-1: none yet) */
    shim_packet *q_head, *q_tail;
    shim_recon  *r_head, *r_tail;
    uint64_t    me_launches;           /* batched ME launches so far (svt_vp9_shim_get_counters) */
    /* geometry and per-stream constants of the stages behind mode decision */
    int         W, H, mi_rows, mi_cols, n_sb;
    size_t      pic_bytes, rec_bytes, coeffs;
    int         q_index, filter_level;
    uint32_t    md_lambda;
    svt_lf_thresh thr;
    svt_vp9_shim_md_callback md_cb;
    void       *md_user;
    void       *h_results, *h_mc, *h_lf;   /* host staging of the callback */
} shim_state;

/* VP9 level limits (max luma picture size, max luma sample rate), indexed like the reference's tables (:109-134) */
static const uint64_t k_max_pic_size[13]    = {36864, 122880, 245760, 552960, 983040, 2228224, 2228224, 8912896, 8912896, 8912896, 35651584, 35651584, 35651584};
static const uint64_t k_max_sample_rate[13] = {552960, 3686400, 7372800, 16588800, 33177600, 66846720, 133693440, 267386880, 534773760,
                                               1069547520ull, 1069547520ull, 2139095040ull, 4278190080ull};

/* ------------------------------------------------------------------------------------------------ */
static void load_defaults(EbSvtVp9EncConfiguration *c) { /* eb_vp9_svt_enc_init_parameter, :1762-1818 */
    c->frame_rate = 30 << 16; c->frame_rate_numerator = 0; c->frame_rate_denominator = 0;
    c->encoder_bit_depth = 8; c->source_width = 0; c->source_height = 0;
    c->qp = 50; c->use_qp_file = 0; c->rate_control_mode = 0; c->target_bit_rate = 7000000;
    c->max_qp_allowed = 63; c->min_qp_allowed = 0; c->base_layer_switch_mode = 0;
    c->enc_mode = 3; c->intra_period = 31; c->pred_structure = 2;
    c->loop_filter = 1; c->use_default_me_hme = 1; c->enable_hme_flag = 1;
    c->search_area_width = 16; c->search_area_height = 7;
    c->profile = 0; c->level = 0;
    c->injector_frame_rate = 60 << 16; c->speed_control_flag = 0;
    c->asm_type = 1;
    c->logical_processors = 0; c->target_socket = -1; c->channel_id = 0; c->active_channel_count = 1;
    c->recon_file = 0;
}

static int level_index(uint32_t level) {
    static const uint32_t ids[13] = {10, 20, 21, 30, 31, 40, 41, 50, 51, 52, 60, 61, 62};
    if (level == 0) return 13; /* decided by the encoder */
    for (int i = 0; i < 13; i++) if (ids[i] == level) return i;
    return 14;
}

/* verify_settings (:2203-2557) on the values copy_api_from_app (:2052-2164) hands it */
static EbErrorType verify(const EbSvtVp9EncConfiguration *in) {
    EbSvtVp9EncConfiguration c = *in;
    if (c.rate_control_mode == 0) { c.max_qp_allowed = 63; c.min_qp_allowed = 0; } /* :2118-2126 */
    int bad = 0;
    const uint32_t W = c.source_width, H = c.source_height;
    const int li = level_index(c.level);
    if (li > 13) bad = 1;
    if (W < 64 || H < 64) bad = 1;
    if (c.pred_structure != 2) bad = 1;
    if ((W % 2) || (H % 2)) bad = 1;
    if (W > 8192 || (W % 8) || H > 4320 || (H % 8)) bad = 1;
    const int res = svt_hip_input_resolution((int32_t)W, (int32_t)H);
    if (res <= 1) { if (c.enc_mode > 9) bad = 1; }
    else if (res == 2) { if (c.enc_mode > 10) bad = 1; }
    else if ((c.enc_mode > 12 && c.tune == 0) || (c.enc_mode > 10 && c.tune >= 1)) bad = 1;
    if (c.qp > 63) bad = 1;
    if (c.intra_period < -2 || c.intra_period > 255) bad = 1;
    if (c.base_layer_switch_mode > 1 || c.loop_filter > 1 || c.use_default_me_hme > 1 || c.enable_hme_flag > 1) bad = 1;
    if (c.search_area_width > 256 || c.search_area_width == 0 || c.search_area_height > 256 || c.search_area_height == 0) bad = 1;
    if (li < 13) {
        if ((uint64_t)W * H > k_max_pic_size[li]) bad = 1;
        if ((uint64_t)c.frame_rate * W * H > (k_max_sample_rate[li] << 16)) bad = 1;
    }
    if (c.frame_rate > (240u << 16) || c.frame_rate == 0) bad = 1;
    if (c.rate_control_mode > 2) bad = 1;
    /* (the "no rate control in the OQ / VMAF tunes" rule is compiled out in the reference: #if !VP9_RC, :2495-2502) */
    if (c.max_qp_allowed > 63) bad = 1;
    else if (c.min_qp_allowed > 62) bad = 1;
    else if (c.min_qp_allowed > c.max_qp_allowed) bad = 1;
    if (c.tune > 2 || c.encoder_bit_depth != 8 || c.profile != 0 || c.speed_control_flag > 1) bad = 1;
    if ((int32_t)c.asm_type < 0 || (int32_t)c.asm_type > 1) bad = 1;
    if (c.target_socket != -1 && c.target_socket != 0 && c.target_socket != 1) bad = 1;
    return bad ? EB_ErrorBadParameter : EB_ErrorNone;
}

static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }
static shim_state *state_of(EbComponentType *h) { return h ? (shim_state *)h->p_component_private : NULL; }

static EbErrorType gpu_fail(shim_state *s) { /* a failed device call ends the stream: every later call reports it (the reference posts
                                                a pipeline error and answers EB_ErrorMax from then on, :437-452, 2914-2917) */
    fprintf(stderr, "SvtVp9Enc (GPU hot path): %s\n", svt_hip_last_error());
    SET_FAILED(s);
    /* direct uploads (SVT_HIP_REGISTER_INPUT=1) read the CALLER's planes: whatever failed, no entry point returns while one may still
       be in flight */
    if (s->register_input)
        for (int k = 0; k < s->n_dev; k++)
            if (s->dev[k].ctx_up) (void)svt_hip_mem_upload_wait(s->dev[k].ctx_up);
    return EB_ErrorMax;
}
#define GPU_TRY(call) do { if ((call) != SVT_HIP_OK) return gpu_fail(s); } while (0)
/* an error that is not the device's, found after work of the group has been enqueued: the stream ends the same way (later calls
 * answer EB_ErrorMax) instead of going on with a group that is half processed */
static EbErrorType stream_fail(shim_state *s, const char *why) {
    fprintf(stderr, "SvtVp9Enc (GPU hot path): %s\n", why);
    SET_FAILED(s);
    return EB_ErrorBadParameter;
}

/* A mode-decision callback's grid, checked on the host before it is uploaded (the pipeline is drained at that point anyway): the
 * rules the device-side driver applies (svt_tq_unit_is_origin: sizes, transform not larger than the block, blocks aligned to their
 * size and inside the picture) and the intra pass's (csrc/intra_kernel.hip: square blocks of 4x4 .. 32x32 with the transform of
 * their own size, modes 0..9).  The device would only flag such a grid and SKIP the offending blocks -- their area would keep the
 * slot's previous picture and be used as a reference.  Returns 1 when the grid is malformed. */
static int grid_malformed(const shim_state *s, const svt_lf_mode_info *g, int intra_picture) {
    for (int ur = 0; ur < s->mi_rows; ur++)
        for (int uc = 0; uc < s->mi_cols; uc++) {
            const int o = svt_tq_unit_is_origin(g, s->mi_cols, s->mi_rows, s->mi_cols, ur, uc);
            if (o < 0) return 1;
            if (!o) continue;
            const svt_lf_mode_info *b = &g[ur * s->mi_cols + uc];
            if (b->is_inter) { if (intra_picture) return 1; continue; }
            const int sub = b->sb_type == 0, w8 = (b->sb_type == 3 || sub) ? 1 : b->sb_type == 6 ? 2 : b->sb_type == 9 ? 4 : 0;
            if (!w8 || b->tx_size != (sub ? 0 : w8 == 1 ? 1 : w8 == 2 ? 2 : 3) || b->pad_[2] > 9) return 1;
            if (sub) { const int m4 = (int)b->pad_[1] | (int)b->pad_[0] << 8; for (int q = 0; q < 4; q++) if (((m4 >> (4 * q)) & 15) > 9) return 1; }
            else if (b->pad_[1] > 9) return 1;
        }
    return 0;
}

/* ------------------------------------------------------------------------------------------------ */
EbErrorType eb_vp9_svt_init_handle(EbComponentType **p_handle, void *p_app_data, EbSvtVp9EncConfiguration *config_ptr) {
    if (!p_handle) return EB_ErrorBadParameter;
    *p_handle = (EbComponentType *)malloc(sizeof(EbComponentType));
    if (!*p_handle) return EB_ErrorInsufficientResources;
    shim_state *s = (shim_state *)calloc(1, sizeof *s);
    if (!s) { free(*p_handle); *p_handle = NULL; return EB_ErrorInsufficientResources; }
    (*p_handle)->n_size = sizeof(EbComponentType);
    (*p_handle)->p_component_private = s;
    (*p_handle)->p_application_private = p_app_data;
    s->last_base = -1;
    if (!config_ptr) return EB_ErrorBadParameter; /* as eb_vp9_svt_enc_init_parameter */
    load_defaults(config_ptr);
    return EB_ErrorNone;
}

EbErrorType eb_vp9_svt_enc_set_parameter(EbComponentType *h, EbSvtVp9EncConfiguration *p) {
    shim_state *s = state_of(h);
    if (!s || !p) return EB_ErrorBadParameter;
    /* copy_api_from_app (:2052-2192) builds the library's copy FIRST -- hierarchical levels, frame rate from numerator /
     * denominator, the automatic intra period -- and verify_settings (:2203) judges that copy */
    EbSvtVp9EncConfiguration c = *p;
    const int levels = (c.tune != 0 && c.rate_control_mode == 0) ? 4 : 3, minigop = 1 << levels;
    if (c.frame_rate_numerator != 0 && c.frame_rate_denominator != 0)
        c.frame_rate = ((c.frame_rate_numerator << 8) / c.frame_rate_denominator) << 8;
    if (c.intra_period == -2) { /* compute_default_intra_period (:2014-2024) */
        const int fps = c.frame_rate < 1000 ? (int)c.frame_rate : (int)(c.frame_rate >> 16);
        const int lo = fps / minigop * minigop, hi = (fps + minigop) / minigop * minigop;
        c.intra_period = abs(fps - hi) > abs(fps - lo) ? lo : hi;
    }
    if (verify(&c) != EB_ErrorNone) return EB_ErrorBadParameter;
    s->cfg = c;
    s->levels = levels;
    s->minigop = minigop;
    s->intra_period = c.intra_period;
    s->configured = 1;
    return EB_ErrorNone;
}

static svt_yuv_planes tight_planes(const struct shim_state *s, uint8_t *base);
static void feeder_stop(shim_dev *d);
static void feeder_start(struct shim_state *s, shim_dev *d);
static EbErrorType join_all(struct shim_state *s);
static void free_dev(shim_state *s, shim_dev *d) {
    feeder_stop(d);
    if (!d->ctx) return;
    if (d->slot) {
        for (int i = 0; i < d->n_slots; i++) {
            shim_slot *t = &d->slot[i];
            svt_hip_mem_free(d->ctx, (void *)t->pa.full.buf);
            svt_hip_mem_free(d->ctx, (void *)t->pa.quarter.buf);
            svt_hip_mem_free(d->ctx, (void *)t->pa.sixteenth.buf);
            void *v[12] = {t->d_rec, t->d_results, t->d_mean, t->d_var, t->d_rcme, t->d_stats, t->d_hist, t->d_mc_mi, t->d_lf_mi, t->d_eob_map, t->d_lfm, t->d_nz};
            for (int k = 0; k < 12; k++) svt_hip_mem_free(d->ctx, v[k]);
        }
        free(d->slot);
        d->slot = NULL;
    }
    void *v[4] = {d->d_src_slab, d->d_pred_slab, d->d_q_slab, d->d_dq_slab};
    for (int k = 0; k < 4; k++) svt_hip_mem_free(d->ctx, v[k]);
    d->d_src_slab = d->d_pred_slab = d->d_q_slab = d->d_dq_slab = NULL;
    while (d->free_recon) { shim_recon *r = d->free_recon; d->free_recon = r->next; svt_hip_host_free(d->ctx, r->host); svt_hip_mem_free(d->ctx, r->d_tight); free(r); }
    if (d->work_key && d->work_key != d->work) svt_hip_encdec_work_destroy(d->ctx_key, d->work_key);
    d->work_key = NULL;
    if (d->work) svt_hip_encdec_work_destroy(d->ctx, d->work);
    d->work = NULL;
    if (d->work_deep) svt_hip_encdec_work_destroy(d->ctx_deep, d->work_deep);
    d->work_deep = NULL;
    if (d->ctx_key && d->ctx_key != d->ctx) svt_hip_ctx_destroy(d->ctx_key);
    d->ctx_key = NULL;
    if (d->ctx_deep && d->ctx_deep != d->ctx) svt_hip_ctx_destroy(d->ctx_deep);
    d->ctx_deep = NULL;
    if (d->ctx_me && d->ctx_me != d->ctx) svt_hip_ctx_destroy(d->ctx_me);
    d->ctx_me = NULL;
    if (d->ctx_in && d->ctx_in != d->ctx) svt_hip_ctx_destroy(d->ctx_in);
    if (d->ctx_up && d->ctx_up != d->ctx && d->ctx_up != d->ctx_in) svt_hip_ctx_destroy(d->ctx_up);
    d->ctx_up = NULL;
    if (d->ctx_out && d->ctx_out != d->ctx) svt_hip_ctx_destroy(d->ctx_out);
    d->ctx_in = d->ctx_out = NULL;
    svt_hip_ctx_destroy(d->ctx);
    d->ctx = NULL;
    (void)s;
}

static int alloc_dev(shim_state *s, shim_dev *d) {
    const int W = s->W, H = s->H;
    const int pad[3] = {68, 32, 16}; /* PA reference paddings, Codec/EbEncHandle.c:1003-1026 */
    /* the mini-GOP being collected, the ones whose work is in flight, and the base picture before them.  Three groups in flight (four
       with the one being collected): the uploads of group g + 2 (16 pictures x ~0.25 ms of PCIe at 4K) cross while the main context
       works on group g + 1 and the deep-layer context on group g -- with one group in flight the input stream waits for the slots of
       the group the device is still coding and the device then waits for the uploads (SVT_HIP_RING_GROUPS, 2..8; 2 = one in flight) */
    {
        const char *rg = getenv("SVT_HIP_RING_GROUPS");
        int         g = rg ? atoi(rg) : 4;
        if (g < 2) g = 2;
        if (g > 8) g = 8;
        d->n_slots = g * s->minigop + 2;
    }
    d->slot = (shim_slot *)calloc((size_t)d->n_slots, sizeof(shim_slot));
    if (!d->slot) return 0;
    const size_t n = (size_t)d->n_slots, units = (size_t)s->mi_rows * s->mi_cols;
    int ok = svt_hip_mem_alloc(d->ctx, n * s->pic_bytes, &d->d_src_slab) == SVT_HIP_OK && svt_hip_mem_alloc(d->ctx, n * s->pic_bytes, &d->d_pred_slab) == SVT_HIP_OK &&
             svt_hip_mem_alloc(d->ctx, n * s->coeffs * sizeof(int16_t), &d->d_q_slab) == SVT_HIP_OK &&
             svt_hip_encdec_work_create(d->ctx, SHIM_WAVE_MAX, W, H, &d->work) == SVT_HIP_OK &&
             (d->ctx_deep == d->ctx || svt_hip_encdec_work_create(d->ctx_deep, SHIM_WAVE_MAX, W, H, &d->work_deep) == SVT_HIP_OK);
    if (ok && d->ctx_key != d->ctx) ok = svt_hip_encdec_work_create(d->ctx_key, 1, W, H, &d->work_key) == SVT_HIP_OK;
    else d->work_key = d->work;
    for (int i = 0; ok && i < d->n_slots; i++) {
        shim_slot *t = &d->slot[i];
        t->number = -1;
        svt_plane *pl[3] = {&t->pa.full, &t->pa.quarter, &t->pa.sixteenth};
        for (int k = 0; ok && k < 3; k++) {
            const int w = W >> k, hh = H >> k;
            void *p = NULL;
            ok = svt_hip_mem_alloc(d->ctx, (size_t)(w + 2 * pad[k]) * (size_t)(hh + 2 * pad[k]), &p) == SVT_HIP_OK;
            pl[k]->buf = (const uint8_t *)p; pl[k]->stride = w + 2 * pad[k]; pl[k]->origin_x = pl[k]->origin_y = pad[k];
            pl[k]->width = w; pl[k]->height = hh;
        }
        t->d_src = (uint8_t *)d->d_src_slab + (size_t)i * s->pic_bytes;
        t->d_pred = (uint8_t *)d->d_pred_slab + (size_t)i * s->pic_bytes;
        t->d_qcoeff = (int16_t *)d->d_q_slab + (size_t)i * s->coeffs;
        t->d_dqcoeff = NULL; /* the encode pass keeps the dequantised coefficients in registers (svt_encdec_picture.d_dqcoeff) */
        void *rec = NULL;
        ok = ok && svt_hip_mem_alloc(d->ctx, s->rec_bytes, &rec) == SVT_HIP_OK &&
             svt_hip_mem_alloc(d->ctx, (size_t)s->n_sb * 85 * sizeof(svt_me_pu_result), &t->d_results) == SVT_HIP_OK &&
             svt_hip_mem_alloc(d->ctx, (size_t)s->n_sb * 85, &t->d_mean) == SVT_HIP_OK &&
             svt_hip_mem_alloc(d->ctx, (size_t)s->n_sb * 85 * sizeof(uint16_t), &t->d_var) == SVT_HIP_OK &&
             svt_hip_mem_alloc(d->ctx, (size_t)s->n_sb * sizeof(uint32_t), &t->d_rcme) == SVT_HIP_OK &&
             svt_hip_mem_alloc(d->ctx, (size_t)s->n_sb * sizeof(svt_me_sb_stats), &t->d_stats) == SVT_HIP_OK &&
             svt_hip_mem_alloc(d->ctx, (2 * SVT_SAD_INTERVALS + 1) * sizeof(uint32_t), &t->d_hist) == SVT_HIP_OK &&
             svt_hip_mem_alloc(d->ctx, units * sizeof(svt_mc_mode_info), &t->d_mc_mi) == SVT_HIP_OK &&
             svt_hip_mem_alloc(d->ctx, units * sizeof(svt_lf_mode_info), &t->d_lf_mi) == SVT_HIP_OK &&
             svt_hip_mem_alloc(d->ctx, (size_t)(W / 4) * (H / 4) * 3 / 2 * sizeof(uint16_t), &t->d_eob_map) == SVT_HIP_OK &&
             svt_hip_mem_alloc(d->ctx, (size_t)s->n_sb * sizeof(svt_lf_mask), &t->d_lfm) == SVT_HIP_OK &&
             svt_hip_mem_alloc(d->ctx, units, &t->d_nz) == SVT_HIP_OK;
        t->d_rec = (uint8_t *)rec;
        t->info.n_sb = (uint32_t)s->n_sb;
    }
    if (ok) { /* take the context's pinned staging buffers now (init is outside the clock of SURVEY 8(d)'s metric, the first pictures are
                 not): one throw-away upload per buffer of the ring */
        uint8_t *z = (uint8_t *)calloc((size_t)W, (size_t)H);
        if (z) {
            const svt_yuv_planes sp = tight_planes(s, d->slot[0].d_src);
            void *const       dd[3] = {sp.y, sp.u, sp.v};
            const void *const ss[3] = {z, z, z};
            const size_t      st3[3] = {(size_t)W, (size_t)W / 2, (size_t)W / 2}, rows[3] = {(size_t)H, (size_t)H / 2, (size_t)H / 2};
            for (int i = 0; i < 4; i++) (void)svt_hip_mem_upload_planes_async(d->ctx_up, 3, dd, st3, ss, st3, st3, rows); /* (a whole picture per buffer) */
            (void)svt_hip_ctx_synchronize(d->ctx_up);
            free(z);
        }
        /* ... and every stream's hardware queue (created by the runtime at the first submission) */
        svt_hip_ctx *all[7] = {d->ctx, d->ctx_in, d->ctx_up, d->ctx_out, d->ctx_key, d->ctx_deep, d->ctx_me};
        const int    scr[7] = {1024, 0, 0, 0, 1024, 256, 256}; /* private segments: the intra pass's kernel (key context; main context: intra blocks of inter pictures), the 32x32 transform */
        for (int i = 0; i < 7; i++) if (all[i]) (void)svt_hip_ctx_warm_scratch(all[i], scr[i]);
        /* ... and the deblocking launches' descriptor buffers at their largest (growing one waits for its stream: the first key frame's intra pass) */
        (void)svt_hip_lf_reserve(d->ctx, SHIM_WAVE_MAX, s->mi_rows, s->mi_cols);
        if (d->ctx_key && d->ctx_key != d->ctx) (void)svt_hip_lf_reserve(d->ctx_key, 1, s->mi_rows, s->mi_cols);
        if (d->ctx_deep && d->ctx_deep != d->ctx) (void)svt_hip_lf_reserve(d->ctx_deep, SHIM_WAVE_MAX, s->mi_rows, s->mi_cols);
    }
    return ok;
}

static void shim_warm_up(const shim_state *real);

EbErrorType eb_vp9_init_encoder(EbComponentType *h) {
    shim_state *s = state_of(h);
    if (!s || !s->configured) return EB_ErrorBadParameter;
    if (s->initialised) return EB_ErrorNone;
    /* target_socket names a CPU socket in the reference (-1 = both); here the GPUs come from SVT_HIP_DEVICES (a comma-separated list
       of ordinals: closed GOPs are dealt round-robin to them) or SVT_HIP_DEVICE (one ordinal, default 0) */
    int         ord[SHIM_MAX_DEV], n = 0;
    const char *list = getenv("SVT_HIP_DEVICES"), *one = getenv("SVT_HIP_DEVICE");
    if (list && *list) {
        for (const char *p = list; *p && n < SHIM_MAX_DEV;) {
            ord[n++] = atoi(p);
            while (*p && *p != ',') p++;
            if (*p == ',') p++;
        }
    }
    if (n == 0) { ord[0] = one ? atoi(one) : 0; n = 1; }
    s->W = (int)s->cfg.source_width; s->H = (int)s->cfg.source_height;
    s->mi_rows = s->H >> 3; s->mi_cols = s->W >> 3;
    s->n_sb = svt_hip_sb_count(s->W, s->H);
    s->pic_bytes = (size_t)s->W * s->H * 3 / 2;
    {
        const size_t pw = (size_t)s->W + 2 * SHIM_REF_PAD, ph = (size_t)s->H + 2 * SHIM_REF_PAD, cpw = (size_t)s->W / 2 + SHIM_REF_PAD, cph = (size_t)s->H / 2 + SHIM_REF_PAD;
        s->rec_bytes = (pw * ph + 2 * cpw * cph + 63) / 64 * 64;
    }
    s->coeffs = (size_t)s->n_sb * SVT_SB_COEFFS;
    /* fixed QP: quantizer_to_qindex[qp] for every picture (the per-layer QP scaling is rate control's, Codec/EbRateControlProcess.c:4581-4735) */
    s->q_index = svt_hip_vp9_qindex_from_qp((int32_t)s->cfg.qp);
    s->filter_level = s->cfg.loop_filter ? svt_hip_lf_level_from_q(svt_hip_vp9_ac_step(s->q_index), 0) : 0;
    s->md_lambda = 4u * (uint32_t)svt_hip_vp9_ac_step(s->q_index);
    svt_hip_lf_thresh_init(&s->thr, 0);
    for (int i = 0; i < n; i++) {
        shim_dev *d = &s->dev[i];
        memset(d, 0, sizeof *d);
        d->ordinal = ord[i];
        /* Streams belong to the DEVICE: a GPU exposes four hardware queues to a process, and the upload / analysis / main / key streams of one
           context take them.  When an ordinal repeats in SVT_HIP_DEVICES -- the N-device host shape exercised on one GPU:
           per-"device" picture ring, workspace, feeder thread -- the later context runs on the streams of the first one of that ordinal instead of
           opening four more (eight streams on four queues made two contexts 18-36 % slower than one; SVT_HIP_SHARE_STREAMS=0 restores that). */
        const shim_dev *peer = NULL;
        { const char *ss = getenv("SVT_HIP_SHARE_STREAMS");
          if (!(ss && atoi(ss) == 0)) for (int k = 0; k < i && !peer; k++) if (s->dev[k].ordinal == ord[i]) peer = &s->dev[k]; }
#define DEV_CTX_CREATE(field) ((peer && peer->field) ? svt_hip_ctx_create_on_stream(&d->field, ord[i], svt_hip_ctx_stream(peer->field)) : svt_hip_ctx_create(&d->field, ord[i]))
        int ok = DEV_CTX_CREATE(ctx) == SVT_HIP_OK;
        if (!ok) { fprintf(stderr, "SvtVp9Enc (GPU hot path): %s\n", svt_hip_last_error()); d->ctx = NULL; }
        if (ok) {
            const char *one = getenv("SVT_HIP_SINGLE_STREAM");
            const char *nu = getenv("SVT_HIP_NO_UPLOAD_STREAM");
            if (one && atoi(one) != 0) d->ctx_in = d->ctx_out = d->ctx_up = d->ctx;
            else {
                if (DEV_CTX_CREATE(ctx_in) != SVT_HIP_OK) { d->ctx_in = NULL; ok = 0; }
                if (ok && nu && atoi(nu) != 0) d->ctx_up = d->ctx_in;
                else if (ok && DEV_CTX_CREATE(ctx_up) != SVT_HIP_OK) { d->ctx_up = NULL; ok = 0; }
                if (ok && !s->cfg.recon_file) d->ctx_out = d->ctx; /* (no reconstruction is fetched: no output stream) */
                else if (ok && DEV_CTX_CREATE(ctx_out) != SVT_HIP_OK) { d->ctx_out = NULL; ok = 0; }
            }
            /* (opt-in: with the upload, analysis and key streams beside the main one the device's four hardware queues are taken; a fifth
               stream shares a queue with one of them and the public-API rate drops -- 3 240 -> 2 540 frames/s on the MI355X box) */
            const char *dp = getenv("SVT_HIP_DEEP_STREAM");
            d->ctx_deep = d->ctx;
            if (ok && !(one && atoi(one) != 0) && dp && atoi(dp) != 0 && DEV_CTX_CREATE(ctx_deep) != SVT_HIP_OK) { d->ctx_deep = NULL; ok = 0; }
            const char *ms = getenv("SVT_HIP_ME_STREAM");
            d->ctx_me = d->ctx;
            /* (on by default when no reconstruction is fetched: with the output stream as well the device's hardware queues are oversubscribed and the
               rate drops -- 2160p enc-mode 3: 689 -> 859 pictures/s without reconstructions, 619 -> 561 with; enc-mode 8: unchanged / 1 515 -> 1 458) */
            const int want_me = ms ? atoi(ms) != 0 : !s->cfg.recon_file;
            if (ok && !(one && atoi(one) != 0) && want_me && !s->md_cb && DEV_CTX_CREATE(ctx_me) != SVT_HIP_OK) { d->ctx_me = NULL; ok = 0; }
            const char *nk = getenv("SVT_HIP_NO_KEY_STREAM");
            d->ctx_key = d->ctx;
            if (ok && !(one && atoi(one) != 0) && !(nk && atoi(nk) != 0) && !s->md_cb && DEV_CTX_CREATE(ctx_key) != SVT_HIP_OK) { d->ctx_key = NULL; ok = 0; }
        }
#undef DEV_CTX_CREATE
        ok = ok && alloc_dev(s, d);
        if (!ok) { /* nothing half-initialised is left behind: the handle is back in its configured state */
            for (int k = i; k >= 0; k--) free_dev(s, &s->dev[k]); /* (reverse: a later context may run on an earlier one's streams) */
            return EB_ErrorInsufficientResources;
        }
    }
    s->n_dev = n;
    s->cur_dev = 0;
    { const char *sg = getenv("SVT_HIP_SPLIT_GOP"); s->split_gop = n > 1 && sg && atoi(sg) != 0; }
    { const char *ri = getenv("SVT_HIP_REGISTER_INPUT"); s->register_input = ri && atoi(ri) != 0; }
    if (s->register_input) svt_hip_host_registry_retain(); /* released in eb_vp9_deinit_encoder: the last encoder out unlocks the application's buffers */
    {   /* feeder threads (SVT_HIP_FEEDER=0 or SVT_HIP_NO_FEEDER=1: off) */
        const char *nf = getenv("SVT_HIP_NO_FEEDER"), *ff = getenv("SVT_HIP_FEEDER"), *one = getenv("SVT_HIP_SINGLE_STREAM");
        const int   want = ff ? atoi(ff) != 0 : 1;
        s->use_feeder = want && !(nf && atoi(nf) != 0) && !(one && atoi(one) != 0) && !s->md_cb && !s->split_gop;
    }
    { const char *pf = getenv("SVT_HIP_SHIM_PROFILE"); s->profile = pf ? atoi(pf) : 0; memset(s->prof_s, 0, sizeof s->prof_s); }
    for (int i = 0; i < n; i++) feeder_start(s, &s->dev[i]);
    if (s->md_cb) {
        s->h_results = malloc((size_t)s->n_sb * 85 * sizeof(svt_me_pu_result));
        s->h_mc = malloc((size_t)s->mi_rows * s->mi_cols * sizeof(svt_mc_mode_info));
        s->h_lf = malloc((size_t)s->mi_rows * s->mi_cols * sizeof(svt_lf_mode_info));
        if (!s->h_results || !s->h_mc || !s->h_lf) { (void)eb_vp9_deinit_encoder(h); return EB_ErrorInsufficientResources; }
    }
    s->initialised = 1;
    if (s->cfg.recon_file) { /* the reconstruction records of two groups, taken now: page-locking 12.4 MB per 4K picture inside the stream cost the first
                                groups ~1 ms per picture (the pool still grows on demand) */
        for (int i = 0; i < n; i++) {
            shim_dev *d = &s->dev[i];
            for (int k = 0; k < 2 * SHIM_MAX_MINIGOP + 2; k++) {
                shim_recon *r = (shim_recon *)calloc(1, sizeof *r);
                void       *hp = NULL, *dp = NULL;
                if (!r || svt_hip_host_alloc(d->ctx_out, s->pic_bytes, &hp) != SVT_HIP_OK || svt_hip_mem_alloc(d->ctx_out, s->pic_bytes, &dp) != SVT_HIP_OK) {
                    if (hp) svt_hip_host_free(d->ctx_out, hp);
                    free(r);
                    break; /* (not an error: reserve_recon allocates what is missing) */
                }
                r->host = (uint8_t *)hp; r->d_tight = (uint8_t *)dp; r->next = d->free_recon; d->free_recon = r;
            }
        }
    }
    shim_warm_up(s);
    return EB_ErrorNone;
}

/* Initialisation is outside the clock of SURVEY 8(d)'s metric (first send_picture -> EOS packet), the first pictures are not: the HIP runtime loads
 * a code object the first time one of its kernels is launched, and the first key frame / first group paid for that inside the stream (~8 ms of a
 * 130-picture run at 4K).  A throw-away encoder of the same preset on a small picture runs one closed GOP's first pictures here -- every kernel of the
 * path has been launched once when the real stream starts.  SVT_HIP_WARMUP=0 skips it. */
static void shim_warm_up(const shim_state *real) {
    static __thread int busy = 0; /* (the throw-away encoder's own init comes through here) */
    const char         *e = getenv("SVT_HIP_WARMUP");
    if (busy || (e && atoi(e) == 0) || real->md_cb) return;
    busy = 1;
    EbComponentType          *h = NULL;
    EbSvtVp9EncConfiguration  cfg;
    const uint32_t            W = 256, H = 192;
    const int                 n = 1 + 2 * 16; /* a key frame and two mini-GOPs */
    uint8_t                  *pic = (uint8_t *)calloc((size_t)W * H * 3 / 2, 1), *rbuf = (uint8_t *)malloc((size_t)W * H * 3 / 2);
    if (pic && rbuf && eb_vp9_svt_init_handle(&h, NULL, &cfg) == EB_ErrorNone) {
        cfg = real->cfg;
        cfg.source_width = W; cfg.source_height = H;
        if (eb_vp9_svt_enc_set_parameter(h, &cfg) == EB_ErrorNone && eb_vp9_init_encoder(h) == EB_ErrorNone) {
            int eos = 0, recon_eos = !cfg.recon_file, guard = 0;
            for (int i = 0; i < n || !(eos && recon_eos); i++) {
                if (i < n) {
                    EbSvtEncInput      in;
                    EbBufferHeaderType b;
                    memset(&in, 0, sizeof in); memset(&b, 0, sizeof b);
                    in.luma = pic; in.cb = pic + (size_t)W * H; in.cr = in.cb + (size_t)W * H / 4;
                    in.y_stride = W; in.cb_stride = in.cr_stride = W / 2;
                    b.size = sizeof b; b.p_buffer = (uint8_t *)&in; b.n_filled_len = W * H * 3 / 2; b.pts = i;
                    b.flags = i == n - 1 ? EB_BUFFERFLAG_EOS : 0;
                    if (eb_vp9_svt_enc_send_picture(h, &b) != EB_ErrorNone) break;
                } else if (++guard > 100000) break;
                for (;;) {
                    EbBufferHeaderType *pk = NULL;
                    if (eb_vp9_svt_get_packet(h, &pk, (uint8_t)(i >= n - 1)) != EB_ErrorNone || !pk) break;
                    eos |= (pk->flags & EB_BUFFERFLAG_EOS) != 0;
                    eb_vp9_svt_release_out_buffer(&pk);
                }
                while (!recon_eos) {
                    EbBufferHeaderType r;
                    memset(&r, 0, sizeof r);
                    r.size = sizeof r; r.p_buffer = rbuf; r.n_alloc_len = W * H * 3 / 2;
                    if (eb_vp9_svt_get_recon(h, &r) != EB_ErrorNone) break;
                    recon_eos = (r.flags & EB_BUFFERFLAG_EOS) != 0;
                }
            }
            (void)eb_vp9_deinit_encoder(h);
        }
        (void)eb_vp9_deinit_handle(h);
    }
    free(pic); free(rbuf);
    busy = 0;
}

EbErrorType eb_vp9_svt_enc_stream_header(EbComponentType *h, EbBufferHeaderType **o) { (void)h; (void)o; return EB_ErrorNone; }
EbErrorType eb_vp9_svt_enc_eos_nal(EbComponentType *h, EbBufferHeaderType **o) { (void)h; (void)o; return EB_ErrorNone; }

/* ------------------------------------------------------------------------------------------------ */
static shim_slot *find_slot(shim_dev *d, int64_t number) {
    for (int i = 0; i < d->n_slots; i++) if (d->slot[i].number == number) return &d->slot[i];
    return NULL;
}
static shim_slot *find_any(shim_state *s, int64_t number, shim_dev **dev) { /* the picture where it was coded, not a handed-over copy */
    for (int k = 0; k < s->n_dev; k++)
        for (int i = 0; i < s->dev[k].n_slots; i++) {
            shim_slot *t = &s->dev[k].slot[i];
            if (t->number == number && !t->is_copy) { if (dev) *dev = &s->dev[k]; return t; }
        }
    return NULL;
}

/* split-GOP mode: the next mini-GOP is coded on device `to`; it predicts from the base picture the current device has just
 * finished -- its analysed planes (motion estimation) and its padded reconstruction (inter prediction) travel device to device,
 * ordered behind the producer's work and in front of the consumer's (svt_hip_ref_handoff_device) */
static EbErrorType handoff_base(shim_state *s, shim_dev *from, shim_dev *to, int64_t number) {
    shim_slot *a = find_slot(from, number);
    if (!a) return EB_ErrorBadParameter;
    shim_slot *b = &to->slot[to->accepted % to->n_slots];
    if (b->has_marker) GPU_TRY(svt_hip_ctx_marker_wait(b->marker_ctx ? b->marker_ctx : to->ctx, b->marker));
    if (b->has_release2) GPU_TRY(svt_hip_ctx_marker_wait(to->ctx_deep, b->release2));
    b->has_release2 = 0; b->marker_ctx = NULL;
    if (b->has_out && to->ctx_out != to->ctx) GPU_TRY(svt_hip_ctx_wait_marker(to->ctx, to->ctx_out, b->out_marker)); /* its old reconstruction may still be on its way out */
    b->has_out = 0;
    const svt_plane *pa[3] = {&a->pa.full, &a->pa.quarter, &a->pa.sixteenth}, *pb[3] = {&b->pa.full, &b->pa.quarter, &b->pa.sixteenth};
    for (int k = 0; k < 3; k++)
        GPU_TRY(svt_hip_ref_handoff_device(from->ctx, pa[k]->buf, to->ctx, (void *)pb[k]->buf, (size_t)pa[k]->stride * (size_t)(pa[k]->height + 2 * pa[k]->origin_y)));
    GPU_TRY(svt_hip_ref_handoff_device(from->ctx, a->d_rec, to->ctx, b->d_rec, s->rec_bytes));
    to->accepted++;
    b->number = number; b->pts = a->pts; b->info = a->info; b->processed = 1; b->coded = 1; b->is_copy = 1;
    GPU_TRY(svt_hip_ctx_marker_record(to->ctx, &b->marker));
    b->has_marker = 1;
    b->release = b->marker; b->has_release = 1;
    return EB_ErrorNone;
}

/* a packet joins the queue (caller's thread only); `ready` = its marker is known already */
static shim_packet *queue_packet(shim_state *s, int64_t pts, uint32_t flags, uint32_t pic_type, int dev, svt_hip_ctx *ctx, uint64_t marker, int ready) {
    shim_packet *p = (shim_packet *)calloc(1, sizeof *p);
    if (!p) return NULL;
    p->hdr.size = sizeof(EbBufferHeaderType);
    p->hdr.pts = p->hdr.dts = pts;
    p->hdr.flags = flags;
    p->hdr.pic_type = pic_type;
    p->hdr.wrapper_ptr = p; /* the round trip of the reference's wrapper_ptr (:2923) */
    p->dev = dev;
    p->ctx = ctx;
    p->marker = marker;
    p->ready = ready;
    if (s->q_tail) s->q_tail->next = p; else s->q_head = p;
    s->q_tail = p;
    return p;
}
static int push_packet(shim_state *s, int64_t pts, uint32_t flags, uint32_t pic_type, int dev, uint64_t marker) {
    return queue_packet(s, pts, flags, pic_type, dev, NULL, marker, 1) ? 0 : -1;
}

/* planes of a slot's tight source / prediction picture and of its padded reference picture */
static svt_yuv_planes tight_planes(const shim_state *s, uint8_t *base) {
    svt_yuv_planes p;
    p.y = base; p.u = base + (size_t)s->W * s->H; p.v = p.u + (size_t)(s->W / 2) * (s->H / 2);
    p.y_stride = s->W; p.uv_stride = s->W / 2; p.width = s->W; p.height = s->H;
    return p;
}
static svt_yuv_planes rec_planes(const shim_state *s, uint8_t *base) {
    const size_t pw = (size_t)s->W + 2 * SHIM_REF_PAD, ph = (size_t)s->H + 2 * SHIM_REF_PAD, cpw = (size_t)s->W / 2 + SHIM_REF_PAD, cph = (size_t)s->H / 2 + SHIM_REF_PAD;
    svt_yuv_planes p;
    p.y = base + SHIM_REF_PAD * pw + SHIM_REF_PAD;
    p.u = base + pw * ph + (SHIM_REF_PAD / 2) * cpw + SHIM_REF_PAD / 2;
    p.v = base + pw * ph + cpw * cph + (SHIM_REF_PAD / 2) * cpw + SHIM_REF_PAD / 2;
    p.y_stride = (int32_t)pw; p.uv_stride = (int32_t)cpw; p.width = s->W; p.height = s->H;
    return p;
}

/* the reconstruction of a picture on its way to eb_vp9_svt_get_recon: W x H luma, then Cb, then Cr (recon_output,
 * Codec/EbEncDecProcess.c:4693-4820), copied to pinned host memory behind the picture's last stage */
/* caller's thread: a record with a pinned buffer joins the reconstruction queue, in coding order; fill_recon completes it */
static shim_recon *reserve_recon(shim_state *s, shim_dev *d, int64_t number) {
    shim_recon *r = d->free_recon;
    if (r) d->free_recon = r->next;
    else {
        r = (shim_recon *)calloc(1, sizeof *r);
        if (!r) return NULL;
        void *hp = NULL;
        if (svt_hip_host_alloc(d->ctx_out, s->pic_bytes, &hp) != SVT_HIP_OK) { free(r); (void)gpu_fail(s); return NULL; }
        r->host = (uint8_t *)hp;
        void *dp = NULL;
        if (svt_hip_mem_alloc(d->ctx_out, s->pic_bytes, &dp) != SVT_HIP_OK) { svt_hip_host_free(d->ctx_out, hp); free(r); (void)gpu_fail(s); return NULL; }
        r->d_tight = (uint8_t *)dp;
    }
    r->dev = (int)(d - s->dev); r->pts = number; r->flags = 0; r->next = NULL; r->ready = 0; r->marker = 0;
    if (s->r_tail) s->r_tail->next = r; else s->r_head = r;
    s->r_tail = r;
    return r;
}
/* enqueuing thread: the copies of picture t into the reserved record */
static EbErrorType fill_recon(shim_state *s, shim_dev *d, shim_slot *t, svt_hip_ctx *coded_on, shim_recon *r) {
    const svt_yuv_planes p = rec_planes(s, t->d_rec);
    const size_t W = (size_t)s->W, H = (size_t)s->H;
    /* the copy runs on the output stream, behind the main stream's work enqueued so far (the picture's deblocking / padding) */
    uint64_t after = 0;
    if ((d->ctx_out != coded_on && (svt_hip_ctx_marker_record(coded_on, &after) != SVT_HIP_OK || svt_hip_ctx_wait_marker(d->ctx_out, coded_on, after) != SVT_HIP_OK)) ||
        /* the interiors of the three padded planes are packed on the device (three strided device-to-device copies at HBM speed) and cross the
           link as ONE linear transfer: three strided device-to-host copies of 2160 / 1080 / 1080 rows each ran at a third of the link's rate
           (public-API path with every reconstruction fetched: 1 270 -> see profiles/r06_api.txt) */
        svt_hip_mem_copy_2d_device(d->ctx_out, r->d_tight, W, p.y, (size_t)p.y_stride, W, H) != SVT_HIP_OK ||
        svt_hip_mem_copy_2d_device(d->ctx_out, r->d_tight + W * H, W / 2, p.u, (size_t)p.uv_stride, W / 2, H / 2) != SVT_HIP_OK ||
        svt_hip_mem_copy_2d_device(d->ctx_out, r->d_tight + W * H + W * H / 4, W / 2, p.v, (size_t)p.uv_stride, W / 2, H / 2) != SVT_HIP_OK ||
        svt_hip_mem_download_2d_async(d->ctx_out, r->host, s->pic_bytes, r->d_tight, s->pic_bytes, s->pic_bytes, 1) != SVT_HIP_OK ||
        svt_hip_ctx_marker_record(d->ctx_out, &r->marker) != SVT_HIP_OK)
        return gpu_fail(s); /* (the record stays in the queue, never ready: the stream has failed) */
    t->out_marker = r->marker; t->has_out = 1;
    __atomic_store_n(&r->ready, 1, __ATOMIC_RELEASE);
    return EB_ErrorNone;
}

/* the parameters motion_estimate_sb reads for this picture, as the reference derives them */
static EbErrorType job_params(shim_state *s, shim_job *j) {
    svt_me_picture_config pc;
    memset(&pc, 0, sizeof pc);
    pc.pic_width = (int32_t)s->cfg.source_width; pc.pic_height = (int32_t)s->cfg.source_height;
    pc.enc_mode = s->cfg.enc_mode; pc.tune = s->cfg.tune;
    /* static_config.frame_rate >> 16, the value the reference's 4K HME widening tests (Codec/EbMotionEstimationProcess.c:55-324) */
    pc.frame_rate = (int32_t)(s->cfg.frame_rate >> 16);
    pc.num_ref_lists = j->n_lists; pc.temporal_layer_index = j->layer; pc.hierarchical_levels = j->levels;
    pc.is_used_as_reference = j->used_as_ref;
    pc.same_ref_poc = j->n_lists == 2 && j->ref0 == j->ref1;
    pc.rate_control_mode = (int32_t)s->cfg.rate_control_mode;
    if (svt_hip_me_params_derive(&j->p, &pc) != SVT_HIP_OK) return EB_ErrorBadParameter;
    if (!s->cfg.use_default_me_hme) { /* eb_vp9_set_me_hme_params_from_config (Codec/EbMotionEstimationProcess.c:316-324) */
        /* verify_settings accepts 1..256 and the reference keeps the value in a uint8_t (256 wraps to 0 there: a configuration
           it cannot run); here the search area is clamped to what the kernel's record holds */
        const uint32_t w = s->cfg.search_area_width, hh = s->cfg.search_area_height;
        j->p.search_area_width  = (uint8_t)(w > 255 ? 255 : w < 1 ? 1 : w);
        j->p.search_area_height = (uint8_t)(hh > 255 ? 255 : hh < 1 ? 1 : hh);
        j->p.enable_hme_flag    = s->cfg.enable_hme_flag;
    }
    return EB_ErrorNone;
}

/* do two parameter sets belong to one launch (svt_hip_me_batch_layers_device)?  The library's own field-by-field rule */
static int same_launch(const svt_me_params *a, const svt_me_params *b) { return svt_hip_me_params_same_launch(a, b) != 0; }

/* the hierarchy between two already-listed pictures lo < hi by bisection, decode order */
static void add_hierarchy(shim_job *jobs, int *n, int64_t lo, int64_t hi, int layer, int levels, int wave0) {
    if (hi - lo < 2) return;
    const int64_t mid = (lo + hi) / 2;
    shim_job *j = &jobs[(*n)++];
    memset(j, 0, sizeof *j);
    j->number = mid; j->ref0 = lo; j->ref1 = hi; j->layer = layer; j->levels = levels; j->n_lists = 2; j->used_as_ref = layer < levels;
    j->wave = wave0 + layer;
    add_hierarchy(jobs, n, lo, mid, layer + 1, levels, wave0);
    add_hierarchy(jobs, n, mid, hi, layer + 1, levels, wave0);
}

/* pred_struct use_subpel_flag of the picture, as the encode pass reads it for its inter prediction (Codec/EbEncDecProcess.c:5507):
 * the ME parameter derivation encodes it as "fractional search off" */
static int job_use_subpel(const shim_job *j) { return j->p.fractional_search_model != 2; }

/* The stages behind mode decision for one batch of mutually independent pictures (a temporal layer of a part of the group, or one
 * picture of a P chain): decision -> svt_hip_encdec_batch_device -> reconstruction output. */
static EbErrorType encode_wave(shim_state *s, shim_dev *d, const shim_job *const *wj, shim_recon *const *wrec, int n, int deep) {
    svt_hip_ctx     *cx = deep ? d->ctx_deep : d->ctx;       /* the context (stream) this wave is enqueued on, and its driver workspace */
    svt_encdec_work *wk = deep ? d->work_deep : d->work;
    svt_encdec_picture pics[SHIM_WAVE_MAX];
    shim_slot         *ts[SHIM_WAVE_MAX];
    svt_encdec_flags   fl;
    {
        svt_encdec_flags_config fc;
        fc.enc_mode = s->cfg.enc_mode; fc.tune = s->cfg.tune; fc.temporal_layer_index = wj[0]->layer; fc.is_used_as_reference = wj[0]->used_as_ref;
        fc.recon_file = (int32_t)s->cfg.recon_file; fc.loop_filter = s->cfg.loop_filter;
        if (svt_hip_encdec_flags_derive(&fc, &fl) != SVT_HIP_OK) return EB_ErrorBadParameter;
    }
    /* fixed-QP mode: the wave's temporal layer scales the sequence QP (QP_SCALING_MODE_0 of the reference's rate-control kernel,
       svt_hip_vp9_layer_qindex); the deblocking level and the stand-in decision's lambda follow the picture's q index */
    const int      q_l = s->cfg.rate_control_mode == 0 ? svt_hip_vp9_layer_qindex((int32_t)s->cfg.qp, s->cfg.tune, wj[0]->levels, wj[0]->layer, 0) : s->q_index;
    const int      level_l = s->cfg.loop_filter ? svt_hip_lf_level_from_q(svt_hip_vp9_ac_step(q_l), 0) : 0;
    const uint32_t lambda_l = 4u * (uint32_t)svt_hip_vp9_ac_step(q_l);
    int n_stand_in = 0;
    int has_intra[SHIM_WAVE_MAX] = {0};
    const svt_me_pu_result *res[SHIM_WAVE_MAX];
    svt_mc_mode_info       *mcs[SHIM_WAVE_MAX];
    svt_lf_mode_info       *lfs[SHIM_WAVE_MAX];
    for (int i = 0; i < n; i++) {
        shim_slot *t = ts[i] = find_slot(d, wj[i]->number);
        shim_slot *r0 = find_slot(d, wj[i]->ref0), *r1 = wj[i]->n_lists == 2 ? find_slot(d, wj[i]->ref1) : r0;
        if (!t || !r0 || !r1) return stream_fail(s, "a picture of the group or one of its references is not resident");
        int decided = 0;
        t->info.decision_source = 0;
        t->info.q_index = q_l; t->info.filter_level = level_l; /* (the callback reads them: the level goes into its grid) */
        if (s->md_cb) { /* the host decides: it needs the ME results, so the pipeline drains here */
            GPU_TRY(svt_hip_mem_download(d->ctx, s->h_results, t->d_results, (size_t)s->n_sb * 85 * sizeof(svt_me_pu_result)));
            memset(s->h_mc, 0, (size_t)s->mi_rows * s->mi_cols * sizeof(svt_mc_mode_info));
            memset(s->h_lf, 0, (size_t)s->mi_rows * s->mi_cols * sizeof(svt_lf_mode_info));
            if (s->md_cb(s->md_user, &t->info, s->h_results, s->h_mc, s->h_lf, s->mi_cols) == 0) {
                if (grid_malformed(s, (const svt_lf_mode_info *)s->h_lf, 0)) return stream_fail(s, "mode-decision callback: malformed mode-info grid");
                GPU_TRY(svt_hip_mem_upload_2d(d->ctx, t->d_mc_mi, (size_t)s->mi_cols * sizeof(svt_mc_mode_info), s->h_mc, (size_t)s->mi_cols * sizeof(svt_mc_mode_info),
                                              (size_t)s->mi_cols * sizeof(svt_mc_mode_info), (size_t)s->mi_rows));
                GPU_TRY(svt_hip_mem_upload_2d(d->ctx, t->d_lf_mi, (size_t)s->mi_cols * sizeof(svt_lf_mode_info), s->h_lf, (size_t)s->mi_cols * sizeof(svt_lf_mode_info),
                                              (size_t)s->mi_cols * sizeof(svt_lf_mode_info), (size_t)s->mi_rows));
                decided = 1;
                t->info.decision_source = 1;
                /* a host decision may hold intra blocks: they go through the intra pass behind the batch */
                const svt_lf_mode_info *g = (const svt_lf_mode_info *)s->h_lf;
                for (size_t u = 0; u < (size_t)s->mi_rows * s->mi_cols && !has_intra[i]; u++) has_intra[i] = !g[u].is_inter;
            }
        }
        if (!decided) { res[n_stand_in] = (const svt_me_pu_result *)t->d_results; mcs[n_stand_in] = (svt_mc_mode_info *)t->d_mc_mi; lfs[n_stand_in] = (svt_lf_mode_info *)t->d_lf_mi; n_stand_in++; }
        svt_encdec_picture *p = &pics[i];
        memset(p, 0, sizeof *p);
        p->d_mc_mi = (const svt_mc_mode_info *)t->d_mc_mi; p->d_lf_mi = (svt_lf_mode_info *)t->d_lf_mi;
        p->src = tight_planes(s, t->d_src); p->pred = tight_planes(s, t->d_pred); p->recon = rec_planes(s, t->d_rec);
        p->ref[0] = rec_planes(s, r0->d_rec); p->ref[1] = rec_planes(s, r1->d_rec);
        p->d_qcoeff = t->d_qcoeff; p->d_dqcoeff = t->d_dqcoeff; p->d_eob_map = (uint16_t *)t->d_eob_map; p->d_lfm = (svt_lf_mask *)t->d_lfm; p->d_nz = (uint8_t *)t->d_nz;
        p->use_subpel = job_use_subpel(wj[i]);
        p->has_intra = has_intra[i];
        t->info.is_used_as_reference = wj[i]->used_as_ref; t->info.do_recon = fl.do_recon; t->info.apply_loop_filter = fl.apply_loop_filter;
        t->info.pad_reference = fl.pad_reference; t->info.q_index = q_l; t->info.filter_level = level_l; t->info.intra_recon_is_source = 0;
    }
    {   /* intra blocks need their neighbours' reconstruction: a wave that would not be reconstructed (the deepest layer without
           reconstructed output) is, when a host decision put intra blocks into it -- the reference reconstructs the SBs that hold them
           (is_intra_sb, Codec/EbEncDecProcess.c:3653-3657); the other flags of the wave stay */
        int any_intra = 0;
        for (int i = 0; i < n; i++) any_intra |= has_intra[i];
        if (any_intra && !fl.do_recon) {
            fl.do_recon = 1;
            for (int i = 0; i < n; i++) ts[i]->info.do_recon = 1;
        }
    }
    if (n_stand_in)
        GPU_TRY(svt_hip_md_default_batch_device(cx, n_stand_in, res, s->W, s->H, lambda_l, level_l, mcs, lfs, s->mi_cols));
    GPU_TRY(svt_hip_encdec_batch_device(cx, wk, n, pics, s->W, s->H, s->mi_cols, q_l, &fl, &s->thr, SHIM_REF_PAD, SHIM_REF_PAD));
    for (int i = 0; i < n; i++) {
        ts[i]->coded = 1;
        if (s->cfg.recon_file) {
            if (!wrec[i]) return stream_fail(s, "reconstruction record");
            const EbErrorType e = fill_recon(s, d, ts[i], cx, wrec[i]);
            if (e != EB_ErrorNone) return e;
        }
    }
    return EB_ErrorNone;
}

/* an intra picture (key frame / intra refresh): the intra encode pass on the GPU (svt_hip_encdec_intra_device: reference samples,
 * predictors, transform / quantisation / reconstruction in coding-dependency order, then deblocking and border).  The host's callback
 * decides its blocks and modes when it wants to (info->is_intra = 1, no ME results); otherwise the stand-in (16x16, DC). */
static EbErrorType encode_intra(shim_state *s, shim_dev *d, shim_slot *t) {
    svt_encdec_flags fl;
    svt_hip_ctx *cx = d->ctx_key; /* (the main context without a key stream) */
    if (d->has_in && d->ctx_in != cx) GPU_TRY(svt_hip_ctx_wait_marker(cx, d->ctx_in, d->in_marker)); /* the picture has been uploaded and analysed */
    {
        svt_encdec_flags_config fc;
        fc.enc_mode = s->cfg.enc_mode; fc.tune = s->cfg.tune; fc.temporal_layer_index = 0; fc.is_used_as_reference = 1;
        fc.recon_file = (int32_t)s->cfg.recon_file; fc.loop_filter = s->cfg.loop_filter;
        if (svt_hip_encdec_flags_derive(&fc, &fl) != SVT_HIP_OK) return EB_ErrorBadParameter;
    }
    const int level = s->cfg.loop_filter ? svt_hip_lf_level_from_q(svt_hip_vp9_ac_step(s->q_index), 1) : 0; /* key-frame rule of eb_vp9_pick_filter_level */
    t->info.is_used_as_reference = 1; t->info.do_recon = fl.do_recon; t->info.apply_loop_filter = fl.apply_loop_filter; t->info.pad_reference = fl.pad_reference;
    t->info.q_index = s->q_index; t->info.filter_level = level; t->info.decision_source = 0; t->info.intra_recon_is_source = 0;
    int decided = 0;
    if (s->md_cb) {
        memset(s->h_mc, 0, (size_t)s->mi_rows * s->mi_cols * sizeof(svt_mc_mode_info));
        memset(s->h_lf, 0, (size_t)s->mi_rows * s->mi_cols * sizeof(svt_lf_mode_info));
        if (s->md_cb(s->md_user, &t->info, NULL, s->h_mc, s->h_lf, s->mi_cols) == 0) {
            if (grid_malformed(s, (const svt_lf_mode_info *)s->h_lf, 1)) return stream_fail(s, "mode-decision callback: malformed mode-info grid of an intra picture");
            GPU_TRY(svt_hip_mem_upload_2d(cx, t->d_lf_mi, (size_t)s->mi_cols * sizeof(svt_lf_mode_info), s->h_lf, (size_t)s->mi_cols * sizeof(svt_lf_mode_info),
                                          (size_t)s->mi_cols * sizeof(svt_lf_mode_info), (size_t)s->mi_rows));
            decided = 1;
            t->info.decision_source = 1;
        }
    }
    const double tk0 = s->profile > 1 ? now_s() : 0.0;
    if (!decided) GPU_TRY(svt_hip_md_intra_default_device(cx, s->W, s->H, level, (svt_lf_mode_info *)t->d_lf_mi, s->mi_cols));
    const double tk1 = s->profile > 1 ? now_s() : 0.0;
    GPU_TRY(svt_hip_mem_set(cx, t->d_mc_mi, 0, (size_t)s->mi_rows * s->mi_cols * sizeof(svt_mc_mode_info))); /* no motion in an intra picture */
    const double tk2 = s->profile > 1 ? now_s() : 0.0;
    svt_encdec_picture p;
    memset(&p, 0, sizeof p);
    p.d_lf_mi = (svt_lf_mode_info *)t->d_lf_mi;
    p.src = tight_planes(s, t->d_src); p.pred = tight_planes(s, t->d_pred); p.recon = rec_planes(s, t->d_rec);
    p.d_qcoeff = t->d_qcoeff; p.d_dqcoeff = t->d_dqcoeff; p.d_eob_map = (uint16_t *)t->d_eob_map; p.d_lfm = (svt_lf_mask *)t->d_lfm; p.d_nz = (uint8_t *)t->d_nz;
    GPU_TRY(svt_hip_encdec_intra_device(cx, d->work_key, &p, s->W, s->H, s->mi_cols, s->q_index, &fl, &s->thr, SHIM_REF_PAD, SHIM_REF_PAD));
    if (s->profile > 1) fprintf(stderr, "SvtVp9Enc key frame %lld: decision %.1f  memset %.1f  intra pass %.1f us\n", (long long)t->number, 1e6 * (tk1 - tk0), 1e6 * (tk2 - tk1), 1e6 * (now_s() - tk2));
    t->coded = 1;
    if (s->cfg.recon_file) {
        shim_recon *r = reserve_recon(s, d, t->number);
        if (!r) return FAILED(s) ? EB_ErrorMax : EB_ErrorInsufficientResources;
        return fill_recon(s, d, t, cx, r);
    }
    return EB_ErrorNone;
}

/* The collected group on the current GOP's device: its pictures are cut into parts as the reference cuts them
 * (svt_hip_minigop_split), all of them go to the GPU's motion estimation in as few launches as their parameter sets allow (one for
 * a regular mini-GOP), followed by the per-SB statistics of every picture and by the stages behind mode decision, one batch per
 * temporal layer.  cut_by_intra: the group was released by an intra refresh -- the reference's pre-assignment buffer then holds
 * the intra picture as its last element (Codec/EbPictureDecisionProcess.c:1641-1646) and the split is made over pending + 1.
 * Nothing here waits for the device (unless the host decides the modes).
 * Two halves: plan_group (caller's thread: the structure of the group, its packets queued in output order) and run_group (the
 * enqueuing: on the device's feeder thread when there is one). */
static EbErrorType plan_group(shim_state *s, int cut_by_intra, int end_of_stream, shim_group *G) {
    shim_dev *d = &s->dev[s->cur_dev];
    shim_job *jobs = G->jobs;
    int       n = 0, n_waves = 0;
    const int64_t first = s->pending_first;
    svt_minigop_part parts[4];
    const int np = svt_hip_minigop_split(s->pending + (cut_by_intra ? 1 : 0), s->levels, cut_by_intra, parts);
    if (np < 1) return EB_ErrorBadParameter;
    if (cut_by_intra) parts[np - 1].length -= 1; /* the intra picture itself is handled by the caller */
    int64_t prev = s->last_base;
    for (int k = 0; k < np; k++) {
        if (parts[k].length < 1) continue;
        const int64_t p0 = first + parts[k].start, base = p0 + parts[k].length - 1;
        if (parts[k].random_access && prev >= 0) { /* base picture first (decode order), then the B hierarchy */
            shim_job *j = &jobs[n++];
            memset(j, 0, sizeof *j);
            j->number = base; j->ref0 = j->ref1 = prev; j->layer = 0; j->levels = parts[k].hierarchical_levels; j->n_lists = 2; j->used_as_ref = 1;
            j->wave = n_waves;
            add_hierarchy(jobs, &n, prev, base, 1, parts[k].hierarchical_levels, n_waves);
            n_waves += parts[k].hierarchical_levels + 1;
        } else { /* low-delay P: the reference's structure tables (Codec/EbPredictionStructure.c) are picture decision, not
                    reproduced -- every picture is predicted from its predecessor and serves as the next one's reference */
            for (int64_t q = p0; q <= base; q++) {
                shim_job *j = &jobs[n++];
                memset(j, 0, sizeof *j);
                j->number = q; j->ref0 = q - 1; j->ref1 = -1; j->layer = 0; j->levels = parts[k].hierarchical_levels; j->n_lists = 1; j->used_as_ref = 1;
                j->wave = n_waves++;
            }
        }
        prev = base;
    }
    for (int i = 0; i < n; i++) {
        const EbErrorType e = job_params(s, &jobs[i]);
        if (e != EB_ErrorNone) return e;
    }
    G->n = n; G->n_waves = n_waves; G->end_of_stream = end_of_stream; G->dev = s->cur_dev;
    G->has_in = d->has_in; G->in_marker = d->in_marker;
    G->has_key = d->has_key && d->ctx_key != d->ctx; G->key_marker = d->key_marker;
    d->has_key = 0; /* (the GOP's later groups follow this one on the main context) */
    G->first = first; G->last = first + s->pending - 1;
    G->oldest = first;
    for (int i = 0; i < n; i++) {
        if (jobs[i].ref0 >= 0 && jobs[i].ref0 < G->oldest) G->oldest = jobs[i].ref0;
        if (jobs[i].n_lists == 2 && jobs[i].ref1 >= 0 && jobs[i].ref1 < G->oldest) G->oldest = jobs[i].ref1;
    }
    /* the group's packets, in decode order; their markers follow when the work has been enqueued */
    for (int i = 0; i < n; i++) {
        shim_slot *t = find_slot(d, jobs[i].number);
        if (!t) return EB_ErrorBadParameter;
        G->pkt[i] = queue_packet(s, t->pts, 0, jobs[i].n_lists == 2 ? 0 /* EB_B_PICTURE */ : 1 /* EB_P_PICTURE */, s->cur_dev, NULL, 0, 0);
        if (!G->pkt[i]) return EB_ErrorInsufficientResources;
        G->rec[i] = NULL;
    }
    /* ... and its reconstructions, in coding order (wave by wave, as they are produced) */
    if (s->cfg.recon_file)
        for (int w = 0; w < n_waves; w++)
            for (int i = 0; i < n; i++)
                if (jobs[i].wave == w) {
                    G->rec[i] = reserve_recon(s, d, jobs[i].number);
                    if (!G->rec[i]) return FAILED(s) ? EB_ErrorMax : EB_ErrorInsufficientResources;
                }
    return EB_ErrorNone;
}

static EbErrorType run_group(shim_state *s, shim_group *G) {
    shim_dev       *d = &s->dev[G->dev];
    const shim_job *jobs = G->jobs;
    const int       n = G->n, n_waves = G->n_waves, end_of_stream = G->end_of_stream;
    /* everything enqueued on the main context from here on sees the group's pictures uploaded and analysed */
    if (G->has_in && d->ctx_in != d->ctx) GPU_TRY(svt_hip_ctx_wait_marker(d->ctx, d->ctx_in, G->in_marker));
    svt_hip_ctx *const cm = (d->ctx_me && !s->split_gop) ? d->ctx_me : d->ctx; /* where the group's search runs */
    if (cm != d->ctx && G->has_in && d->ctx_in != cm) GPU_TRY(svt_hip_ctx_wait_marker(cm, d->ctx_in, G->in_marker));
    /* launches: classes of pictures whose parameter sets may share one */
    int done[SHIM_MAX_MINIGOP] = {0};
    for (int i = 0; i < n; i++) {
        if (done[i]) continue;
        svt_pa_picture    cur[SHIM_MAX_MINIGOP], r0[SHIM_MAX_MINIGOP], r1[SHIM_MAX_MINIGOP];
        svt_me_params     pp[SHIM_MAX_MINIGOP];
        svt_me_pu_result *res[SHIM_MAX_MINIGOP];
        uint32_t         *rc[SHIM_MAX_MINIGOP];
        int               m = 0;
        for (int k = i; k < n; k++) {
            if (done[k] || !same_launch(&jobs[i].p, &jobs[k].p)) continue;
            shim_slot *t = find_slot(d, jobs[k].number), *a = find_slot(d, jobs[k].ref0), *b = find_slot(d, jobs[k].n_lists == 2 ? jobs[k].ref1 : jobs[k].ref0);
            if (!t || !a || !b) return stream_fail(s, "a picture of the group or one of its references is not resident");
            cur[m] = t->pa; r0[m] = a->pa; r1[m] = b->pa;
            pp[m] = jobs[k].p; res[m] = (svt_me_pu_result *)t->d_results; rc[m] = (uint32_t *)t->d_rcme;
            done[k] = 1;
            m++;
        }
        GPU_TRY(svt_hip_me_batch_layers_device(cm, m, cur, r0, r1, pp, res, s->cfg.rate_control_mode ? rc : NULL));
        __atomic_add_fetch(&s->me_launches, 1, __ATOMIC_RELAXED);
    }
    /* the tail of the ME kernel process per picture (Codec/EbMotionEstimationProcess.c:1047-1237): stationary-edge flags and the
       rate-control histograms from the ME results and the picture-analysis variances, all device resident */
    const int res_class = svt_hip_input_resolution((int32_t)s->cfg.source_width, (int32_t)s->cfg.source_height);
    for (int i = 0; i < n; i++) {
        shim_slot *t = find_slot(d, jobs[i].number);
        svt_me_sb_stats_params sp;
        memset(&sp, 0, sizeof sp);
        sp.pic_width = (int32_t)s->cfg.source_width; sp.pic_height = (int32_t)s->cfg.source_height; sp.input_resolution = res_class;
        sp.temporal_layer_index = jobs[i].layer; sp.slice_type = jobs[i].n_lists == 2 ? 0 : 1;
        sp.run_part2 = !end_of_stream; sp.rate_control_mode = (int32_t)s->cfg.rate_control_mode;
        GPU_TRY(svt_hip_mem_set(cm, t->d_hist, 0, (2 * SVT_SAD_INTERVALS + 1) * sizeof(uint32_t)));
        GPU_TRY(svt_hip_me_sb_stats_device(cm, &sp, (const svt_me_pu_result *)t->d_results, (const uint16_t *)t->d_var,
                                           s->cfg.rate_control_mode ? (const uint32_t *)t->d_rcme : NULL, (svt_me_sb_stats *)t->d_stats, (uint32_t *)t->d_hist,
                                           (uint32_t *)t->d_hist + 2 * SVT_SAD_INTERVALS));
        t->info.is_intra = 0; t->info.temporal_layer_index = jobs[i].layer; t->info.hierarchical_levels = jobs[i].levels;
        t->info.num_ref_lists = jobs[i].n_lists;
        t->info.ref_picture_number[0] = jobs[i].ref0; t->info.ref_picture_number[1] = jobs[i].n_lists == 2 ? jobs[i].ref1 : -1;
        t->info.device_ordinal = d->ordinal;
    }
    uint64_t me_marker = 0;
    GPU_TRY(svt_hip_ctx_marker_record(cm, &me_marker));
    for (int i = 0; i < n; i++) { shim_slot *t = find_slot(d, jobs[i].number); t->processed = 1; t->has_marker = 1; t->marker = me_marker; t->marker_ctx = cm; }
    if (cm != d->ctx) GPU_TRY(svt_hip_ctx_wait_marker(d->ctx, cm, me_marker)); /* the layers below read the search's results */
    /* the waves predict from the GOP's intra picture (directly or through pictures that did): behind its encode pass on the key context */
    if (G->has_key) GPU_TRY(svt_hip_ctx_wait_marker(d->ctx, d->ctx_key, G->key_marker));
    /* the stages behind mode decision, wave by wave (a wave = the pictures of one temporal layer of one part: their references
       belong to earlier waves) */
    uint64_t     wave_marker[SHIM_MAX_MINIGOP + 8];
    svt_hip_ctx *wave_ctx[SHIM_MAX_MINIGOP + 8];
    /* the waves of the two deepest layers of a part (temporal layer >= 2 and within one of the part's deepest) go to the deep-layer
       context, behind everything the main context holds at that point (the part's shallower waves: their references).  With a host
       decision callback the pipeline drains per picture anyway: everything stays on the main context. */
    int part_deepest[SHIM_MAX_MINIGOP + 8], any_deep = 0;
    for (int w = 0; w < n_waves; w++) part_deepest[w] = 0;
    for (int w = 0; w < n_waves; w++) { /* waves of a part are consecutive, layer 0 first: a part's deepest layer is its last wave's */
        int lw = -1;
        for (int i = 0; i < n; i++) if (jobs[i].wave == w) lw = jobs[i].layer;
        for (int v = w; v >= 0; v--) { /* back to the part's first wave (the latest wave of layer 0 at or before w) */
            if (lw > part_deepest[v]) part_deepest[v] = lw;
            int lv = -1;
            for (int i = 0; i < n; i++) if (jobs[i].wave == v) lv = jobs[i].layer;
            if (lv == 0) break;
        }
    }
    for (int w = 0; w < n_waves; w++) {
        const shim_job *wj[SHIM_MAX_MINIGOP];
        shim_recon     *wrec[SHIM_MAX_MINIGOP];
        int             m = 0;
        for (int i = 0; i < n; i++) if (jobs[i].wave == w) { wj[m] = &jobs[i]; wrec[m] = G->rec[i]; m++; }
        const int deep = m && d->ctx_deep != d->ctx && !s->md_cb && wj[0]->layer >= 2 && wj[0]->layer >= part_deepest[w] - 1;
        if (deep) {
            uint64_t behind = 0;
            GPU_TRY(svt_hip_ctx_marker_record(d->ctx, &behind));
            GPU_TRY(svt_hip_ctx_wait_marker(d->ctx_deep, d->ctx, behind));
            any_deep = 1;
        }
        for (int b = 0; b < m; b += SHIM_WAVE_MAX) {
            const EbErrorType e = encode_wave(s, d, wj + b, wrec + b, m - b < SHIM_WAVE_MAX ? m - b : SHIM_WAVE_MAX, deep);
            if (e != EB_ErrorNone) return e;
        }
        wave_marker[w] = 0;
        wave_ctx[w] = deep ? d->ctx_deep : d->ctx;
        GPU_TRY(svt_hip_ctx_marker_record(wave_ctx[w], &wave_marker[w]));
    }
    for (int i = 0; i < n; i++) { /* the packets (queued in decode order by plan_group) learn what they wait for */
        shim_slot *t = find_slot(d, jobs[i].number);
        t->marker = wave_marker[jobs[i].wave];
        t->marker_ctx = wave_ctx[jobs[i].wave];
        G->pkt[i]->ctx = t->marker_ctx; G->pkt[i]->marker = t->marker;
        __atomic_store_n(&G->pkt[i]->ready, 1, __ATOMIC_RELEASE);
    }
    /* from this point of the main stream on nothing enqueued so far reads the group's pictures or the pictures it predicted from: the
       input side may overwrite their slots behind it */
    uint64_t end_marker = 0, deep_end = 0;
    GPU_TRY(svt_hip_ctx_marker_record(d->ctx, &end_marker));
    if (any_deep) GPU_TRY(svt_hip_ctx_marker_record(d->ctx_deep, &deep_end)); /* ... and the deep-layer context's readers */
    for (int i = 0; i < n; i++) {
        const int64_t who[3] = {jobs[i].number, jobs[i].ref0, jobs[i].n_lists == 2 ? jobs[i].ref1 : -1};
        for (int k = 0; k < 3; k++) {
            shim_slot *t = who[k] >= 0 ? find_slot(d, who[k]) : NULL;
            if (t) { t->release = end_marker; t->has_release = 1; if (any_deep) { t->release2 = deep_end; t->has_release2 = 1; } }
        }
    }
    return EB_ErrorNone;
}

/* ---- the device's feeder thread ---- */
static void *feeder_main(void *arg) {
    shim_dev *d = (shim_dev *)arg;
    pthread_mutex_lock(&d->mu);
    for (;;) {
        while (d->job_state != 1 && !d->stop) pthread_cond_wait(&d->cv, &d->mu);
        if (d->stop) break;
        d->job_state = 2;
        pthread_mutex_unlock(&d->mu);
        const EbErrorType e = run_group(d->owner, &d->group);
        pthread_mutex_lock(&d->mu);
        d->job_result = (int)e;
        d->job_state = 0;
        pthread_cond_broadcast(&d->cv);
    }
    pthread_mutex_unlock(&d->mu);
    return NULL;
}
/* the caller waits for the group it posted to this device; a failed group ends the stream here */
static EbErrorType dev_join(shim_state *s, shim_dev *d) {
    if (!d->outstanding) return EB_ErrorNone;
    pthread_mutex_lock(&d->mu);
    while (d->job_state != 0) pthread_cond_wait(&d->cv, &d->mu);
    const EbErrorType e = (EbErrorType)d->job_result;
    d->job_result = (int)EB_ErrorNone;
    pthread_mutex_unlock(&d->mu);
    d->outstanding = 0;
    if (e != EB_ErrorNone) { SET_FAILED(s); return e; }
    return EB_ErrorNone;
}
static EbErrorType join_all(shim_state *s) {
    EbErrorType r = EB_ErrorNone;
    for (int k = 0; k < s->n_dev; k++) { const EbErrorType e = dev_join(s, &s->dev[k]); if (e != EB_ErrorNone) r = e; }
    return r;
}
static void feeder_start(shim_state *s, shim_dev *d) {
    d->owner = s;
    pthread_mutex_init(&d->mu, NULL);
    pthread_cond_init(&d->cv, NULL);
    d->job_state = 0; d->stop = 0; d->outstanding = 0; d->job_result = (int)EB_ErrorNone;
    d->has_feeder = s->use_feeder && pthread_create(&d->feeder, NULL, feeder_main, d) == 0;
}
static void feeder_stop(shim_dev *d) {
    if (!d->owner) return;
    if (d->has_feeder) {
        pthread_mutex_lock(&d->mu);
        while (d->job_state != 0) pthread_cond_wait(&d->cv, &d->mu);
        d->stop = 1;
        pthread_cond_broadcast(&d->cv);
        pthread_mutex_unlock(&d->mu);
        pthread_join(d->feeder, NULL);
        d->has_feeder = 0;
    }
    pthread_mutex_destroy(&d->mu);
    pthread_cond_destroy(&d->cv);
    d->owner = NULL;
}

static EbErrorType flush_pending(shim_state *s, int cut_by_intra, int end_of_stream) {
    if (!s->pending) return EB_ErrorNone;
    shim_dev *d = &s->dev[s->cur_dev];
    { const EbErrorType e = dev_join(s, d); if (e != EB_ErrorNone) return e; } /* one group per device at a time */
    shim_group *G = &d->group;
    { const EbErrorType e = plan_group(s, cut_by_intra, end_of_stream, G); if (e != EB_ErrorNone) { SET_FAILED(s); return e; } }
    s->last_base = G->last;
    s->pending = 0;
    /* a regular group goes to the device's feeder; what the caller follows up at once (the intra picture behind a cut group, the
       end of the stream, a hand-over to another device) is enqueued here */
    if (d->has_feeder && !cut_by_intra && !end_of_stream && !s->split_gop) {
        pthread_mutex_lock(&d->mu);
        d->job_state = 1;
        pthread_cond_broadcast(&d->cv);
        pthread_mutex_unlock(&d->mu);
        d->outstanding = 1;
        return EB_ErrorNone;
    }
    { const EbErrorType e = run_group(s, G); if (e != EB_ErrorNone) { SET_FAILED(s); return e; } }
    if (s->split_gop && !cut_by_intra && !end_of_stream) { /* the next mini-GOP of this GOP goes to the next device */
        const int next = (s->cur_dev + 1) % s->n_dev;
        const EbErrorType e = handoff_base(s, d, &s->dev[next], s->last_base);
        if (e != EB_ErrorNone) return e;
        s->cur_dev = next;
    }
    return EB_ErrorNone;
}

EbErrorType eb_vp9_svt_enc_send_picture(EbComponentType *h, EbBufferHeaderType *b) {
    shim_state *s = state_of(h);
    if (!s || !s->initialised) return EB_ErrorBadParameter;
    if (FAILED(s)) return EB_ErrorMax;
    if (s->eos) return EB_ErrorBadParameter;
    const int end = !b || !b->p_buffer || (b->flags & EB_BUFFERFLAG_EOS);
    EbErrorType e = EB_ErrorNone;
    const double t_enter = s->profile ? now_s() : 0.0;
    double       tp = t_enter;
#define PROF(k) do { if (s->profile) { const double n_ = now_s(); s->prof_s[k] += n_ - tp; tp = n_; } } while (0)
    if (b && b->p_buffer) {
        const EbSvtEncInput *in = (const EbSvtEncInput *)b->p_buffer;
        if (!in->luma || in->y_stride < s->cfg.source_width) return EB_ErrorBadParameter;
        if ((in->cb && in->cb_stride < s->cfg.source_width / 2) || (in->cr && in->cr_stride < s->cfg.source_width / 2)) return EB_ErrorBadParameter;
        const int64_t n = s->next_number;
        const int     W = s->W, H = s->H;
        const int     intra = n == 0 || (s->intra_period >= 0 && n % (s->intra_period + 1) == 0);
        if (intra) { /* an intra refresh closes the GOP: what is waiting is coded on its own device, cut as the reference cuts it */
            if ((e = flush_pending(s, 1, 0)) != EB_ErrorNone) return e;
            if (n) { s->gop++; if (!s->split_gop) s->cur_dev = svt_hip_gop_owner(s->gop, s->n_dev); }
            s->last_base = -1;
        }
        PROF(3);
        shim_dev  *d = &s->dev[s->cur_dev];
        shim_slot *t = &d->slot[d->accepted % d->n_slots];
        /* the group the device's feeder is enqueuing may still write this slot's markers (never with the ring's regular distance of two
           mini-GOPs + 2; short GOPs on several devices can come closer): wait for it */
        if (d->outstanding && t->number >= d->group.oldest) { if ((e = dev_join(s, d)) != EB_ErrorNone) return e; }
        /* the slot's previous picture (2 mini-GOPs + 2 ago on this device) must have left the GPU before its buffers are overwritten:
           the input stream waits for the main stream's marker behind its last reader (a device-side wait; the host blocks only in
           the staging ring of the upload below, as the reference blocks when its picture pool is exhausted) */
        if (t->has_release) GPU_TRY(svt_hip_ctx_wait_marker(d->ctx_up, d->ctx, t->release));
        else if (t->has_marker) GPU_TRY(svt_hip_ctx_wait_marker(d->ctx_up, t->marker_ctx ? t->marker_ctx : d->ctx, t->marker));
        if (t->has_release2) GPU_TRY(svt_hip_ctx_wait_marker(d->ctx_up, d->ctx_deep, t->release2));
        if (t->has_out && d->ctx_out != d->ctx_up) GPU_TRY(svt_hip_ctx_wait_marker(d->ctx_up, d->ctx_out, t->out_marker));
        /* the copy the reference makes in copy_frame_buffer (:2743-2796), into pinned staging: the caller's planes are free again
           on return; the transfer and the analysis below run asynchronously */
        const svt_yuv_planes sp = tight_planes(s, t->d_src);
        /* SVT_HIP_REGISTER_INPUT=1: the application promises that its input buffers stay allocated while the encoder lives (a fixed pool,
           as the reference's application has): they are page-locked on first sight and read by the DMA engines directly, without the
           staging copy (svt_hip_mem_upload_2d_direct) */
        PROF(0);
        if (!s->register_input && in->cb && in->cr) { /* the three planes through one staging slot, one copy to the device (Y | Cb | Cr are back to back there) */
            void *const       dd[3] = {sp.y, sp.u, sp.v};
            const void *const ss[3] = {in->luma, in->cb, in->cr};
            const size_t      dst_st[3] = {(size_t)W, (size_t)W / 2, (size_t)W / 2}, src_st[3] = {in->y_stride, in->cb_stride, in->cr_stride};
            const size_t      wb[3] = {(size_t)W, (size_t)W / 2, (size_t)W / 2}, rows[3] = {(size_t)H, (size_t)H / 2, (size_t)H / 2};
            GPU_TRY(svt_hip_mem_upload_planes_async(d->ctx_up, 3, dd, dst_st, ss, src_st, wb, rows));
        } else {
        int32_t (*up)(svt_hip_ctx *, void *, size_t, const void *, size_t, size_t, size_t) = s->register_input ? svt_hip_mem_upload_2d_direct : svt_hip_mem_upload_2d_async;
        GPU_TRY(up(d->ctx_up, sp.y, (size_t)W, in->luma, in->y_stride, (size_t)W, (size_t)H));
        if (in->cb) GPU_TRY(up(d->ctx_up, sp.u, (size_t)W / 2, in->cb, in->cb_stride, (size_t)W / 2, (size_t)H / 2));
        else GPU_TRY(svt_hip_mem_set(d->ctx_up, sp.u, 128, (size_t)(W / 2) * (H / 2)));
        if (in->cr) GPU_TRY(up(d->ctx_up, sp.v, (size_t)W / 2, in->cr, in->cr_stride, (size_t)W / 2, (size_t)H / 2));
        else GPU_TRY(svt_hip_mem_set(d->ctx_up, sp.v, 128, (size_t)(W / 2) * (H / 2)));
        }
        PROF(1);
        const uint8_t *lum = sp.y;
        const int32_t  stride = W;
        if (d->ctx_up != d->ctx_in) { /* the analysis follows the picture's copy (and, through it, the slot's release) */
            uint64_t up_marker = 0;
            GPU_TRY(svt_hip_ctx_marker_record(d->ctx_up, &up_marker));
            GPU_TRY(svt_hip_ctx_wait_marker(d->ctx_in, d->ctx_up, up_marker));
        }
        GPU_TRY(svt_hip_pa_prepare_batch_device(d->ctx_in, 1, &lum, &stride, &t->pa, 1));
        GPU_TRY(svt_hip_pa_mean_variance_device(d->ctx_in, &t->pa.full, (uint8_t *)t->d_mean, (uint16_t *)t->d_var));
        GPU_TRY(svt_hip_ctx_marker_record(d->ctx_in, &d->in_marker));
        d->has_in = 1;
        /* direct uploads: the caller's planes have been read when this returns (the analysis kernels above are enqueued behind the copies
           and are not waited for) */
        if (s->register_input) GPU_TRY(svt_hip_mem_upload_wait(d->ctx_up));
        PROF(2);
        /* the picture is accepted from here on */
        s->next_number = n + 1;
        d->accepted++;
        t->number = n; t->pts = b->pts; t->processed = 0; t->coded = 0; t->has_marker = 0; t->has_release = 0; t->has_release2 = 0; t->marker_ctx = NULL; t->has_out = 0; t->is_copy = 0;
        memset(&t->info, 0, sizeof t->info);
        t->info.picture_number = (uint64_t)n; t->info.n_sb = (uint32_t)s->n_sb; t->info.device_ordinal = d->ordinal;
        if (intra) {
            t->info.is_intra = 1; t->info.num_ref_lists = 0; t->info.temporal_layer_index = 0; t->info.hierarchical_levels = s->levels;
            t->info.ref_picture_number[0] = t->info.ref_picture_number[1] = -1;
            /* the intra pass is enqueued here: on the key context (nothing the device's feeder touches -- unless reconstructions are
               fetched: the output context is shared), or on the device's main context */
            if (d->ctx_key == d->ctx || s->cfg.recon_file) { if ((e = dev_join(s, d)) != EB_ErrorNone) return e; }
            if ((e = encode_intra(s, d, t)) != EB_ErrorNone) return e;
            GPU_TRY(svt_hip_ctx_marker_record(d->ctx_key, &t->marker));
            t->has_marker = 1; t->processed = 1; t->marker_ctx = d->ctx_key;
            if (d->ctx_key == d->ctx) { t->release = t->marker; t->has_release = 1; }
            d->key_marker = t->marker; d->has_key = 1;
            if (!queue_packet(s, t->pts, 0, 2 /* EB_I_PICTURE */, s->cur_dev, d->ctx_key, t->marker, 1)) return EB_ErrorInsufficientResources;
            s->last_base = n;
            if (s->profile > 1) fprintf(stderr, "SvtVp9Enc key frame %lld: %.1f us on the caller's thread\n", (long long)n, 1e6 * (now_s() - tp));
            PROF(4);
        } else {
            if (!s->pending) s->pending_first = n;
            if (++s->pending == s->minigop) e = flush_pending(s, 0, 0);
            PROF(3);
        }
    }
    if (end && e == EB_ErrorNone) {
        e = flush_pending(s, 0, 1);
        s->eos = 1;
        if (e == EB_ErrorNone) {
            if (s->q_tail) s->q_tail->hdr.flags |= EB_BUFFERFLAG_EOS; /* the last picture's packet closes the stream */
            else {
                uint64_t m = 0;
                GPU_TRY(svt_hip_ctx_marker_record(s->dev[s->cur_dev].ctx, &m));
                if (push_packet(s, b ? b->pts : 0, EB_BUFFERFLAG_EOS, 0, s->cur_dev, m)) e = EB_ErrorInsufficientResources;
            }
            if (s->r_tail) s->r_tail->flags |= EB_BUFFERFLAG_EOS; /* ... and the last reconstruction the recon stream (recon_output :4711-4713) */
        }
    }
    if (s->profile) s->prof_s[5] += now_s() - t_enter;
#undef PROF
    return e;
}

/* Non-blocking while pictures are still being sent (EB_NoErrorEmptyQueue when the next packet's GPU work has not finished);
 * with pic_send_done the call waits for it, as eb_vp9_svt_get_packet does (:2880-2915: eb_vp9_get_full_object vs the
 * non-blocking variant). */
EbErrorType eb_vp9_svt_get_packet(EbComponentType *h, EbBufferHeaderType **p_buffer, uint8_t pic_send_done) {
    shim_state *s = state_of(h);
    if (!s || !p_buffer) return EB_ErrorBadParameter;
    if (FAILED(s)) return EB_ErrorMax;
    shim_packet *p = s->q_head;
    if (!p) return EB_NoErrorEmptyQueue;
    /* the newest packet stays in the queue until the library knows whether it is the last one (it then carries EB_BUFFERFLAG_EOS): the
       end-of-stream buffer arrives in a send_picture call of its own (App/EbAppProcessCmd.c:483-494) */
    if (p == s->q_tail && !s->eos) return EB_NoErrorEmptyQueue;
    if (!__atomic_load_n(&p->ready, __ATOMIC_ACQUIRE)) { /* its group is still with the device's feeder */
        if (!pic_send_done) return EB_NoErrorEmptyQueue;
        const EbErrorType je = dev_join(s, &s->dev[p->dev]);
        if (je != EB_ErrorNone) return je;
        if (!__atomic_load_n(&p->ready, __ATOMIC_ACQUIRE)) return EB_ErrorMax;
    }
    svt_hip_ctx *ctx = p->ctx ? p->ctx : s->dev[p->dev].ctx;
    if (ctx) {
        if (pic_send_done) { GPU_TRY(svt_hip_ctx_marker_wait(ctx, p->marker)); }
        else {
            const int32_t q = svt_hip_ctx_marker_query(ctx, p->marker);
            if (q < 0) return gpu_fail(s);
            if (q == 0) return EB_NoErrorEmptyQueue;
        }
    }
    s->q_head = p->next;
    if (!s->q_head) s->q_tail = NULL;
    p->next = NULL;
    *p_buffer = &p->hdr;
    return EB_ErrorNone;
}

void eb_vp9_svt_release_out_buffer(EbBufferHeaderType **p_buffer) {
    if (p_buffer && *p_buffer && (*p_buffer)->wrapper_ptr) {
        free((*p_buffer)->wrapper_ptr);
        *p_buffer = NULL;
    }
}

/* eb_vp9_svt_get_recon (:2837-2865): non-blocking; the next reconstructed picture, copied into the caller's p_buffer (the caller
 * allocates W * H * 3 / 2 bytes, App/EbAppContext.c allocate_output_recon_buffers), with the header fields copy_output_recon_buffer
 * copies (:2815-2830) */
EbErrorType eb_vp9_svt_get_recon(EbComponentType *h, EbBufferHeaderType *p_buffer) {
    shim_state *s = state_of(h);
    if (!s) return EB_ErrorBadParameter;
    if (!s->cfg.recon_file) return EB_ErrorMax; /* recon is not enabled */
    if (FAILED(s)) return EB_ErrorMax;
    shim_recon *r = s->r_head;
    if (!r || !p_buffer) return EB_NoErrorEmptyQueue;
    if (r == s->r_tail && !s->eos) return EB_NoErrorEmptyQueue; /* as for packets: the last reconstruction carries EB_BUFFERFLAG_EOS */
    shim_dev *d = &s->dev[r->dev];
    if (!__atomic_load_n(&r->ready, __ATOMIC_ACQUIRE)) return EB_NoErrorEmptyQueue; /* its group is still with the device's feeder */
    const int32_t q = svt_hip_ctx_marker_query(d->ctx_out, r->marker);
    if (q < 0) return gpu_fail(s);
    if (q == 0) return EB_NoErrorEmptyQueue;
    if (p_buffer->p_buffer) {
        if (p_buffer->n_alloc_len && p_buffer->n_alloc_len < s->pic_bytes) return EB_ErrorBadParameter;
        /* Y | Cb | Cr, tight: W x (H * 3 / 2) bytes as rows of W for the library's copy pool (a few threads) */
        svt_copy_rows_mt((uint8_t *)p_buffer->p_buffer, (size_t)s->W, r->host, (size_t)s->W, (size_t)s->W, (size_t)s->H * 3 / 2);
    }
    p_buffer->size = sizeof(EbBufferHeaderType);
    p_buffer->n_filled_len = (uint32_t)s->pic_bytes;
    p_buffer->pts = r->pts; p_buffer->dts = 0;
    p_buffer->flags = r->flags;
    p_buffer->pic_type = 0;
    s->r_head = r->next;
    if (!s->r_head) s->r_tail = NULL;
    r->next = d->free_recon;
    d->free_recon = r;
    return EB_ErrorNone;
}

EbErrorType eb_vp9_deinit_encoder(EbComponentType *h) {
    shim_state *s = state_of(h);
    if (!s) return EB_ErrorNone; /* the reference accepts a NULL component here (:1846) */
    (void)join_all(s);
    while (s->q_head) { shim_packet *p = s->q_head; s->q_head = p->next; free(p); }
    s->q_tail = NULL;
    while (s->r_head) { /* undelivered reconstructions go back to their device's pool, which is freed with the device */
        shim_recon *r = s->r_head;
        s->r_head = r->next;
        r->next = s->dev[r->dev].free_recon;
        s->dev[r->dev].free_recon = r;
    }
    s->r_tail = NULL;
    for (int k = s->n_dev - 1; k >= 0; k--) free_dev(s, &s->dev[k]); /* (reverse: a later context may run on an earlier one's streams) */
    s->n_dev = 0;
    /* every stream has drained (join_all, free_dev): nothing reads the application's input buffers any more -- the page locks this encoder
       took on them go with it (the registry is process-wide and counted: the last encoder that leaves unlocks) */
    if (s->register_input) { svt_hip_host_registry_release(); s->register_input = 0; }
    if (s->profile && s->next_number)
        fprintf(stderr, "SvtVp9Enc host time per picture (us): slot wait %.1f  upload %.1f  analysis %.1f  group hand-over %.1f  intra %.1f  | send_picture %.1f  (%lld pictures)\n",
                1e6 * s->prof_s[0] / (double)s->next_number, 1e6 * s->prof_s[1] / (double)s->next_number, 1e6 * s->prof_s[2] / (double)s->next_number,
                1e6 * s->prof_s[3] / (double)s->next_number, 1e6 * s->prof_s[4] / (double)s->next_number, 1e6 * s->prof_s[5] / (double)s->next_number, (long long)s->next_number);
    free(s->h_results); free(s->h_mc); free(s->h_lf);
    s->h_results = s->h_mc = s->h_lf = NULL;
    s->initialised = 0;
    return EB_ErrorNone;
}

EbErrorType eb_vp9_deinit_handle(EbComponentType *h) {
    if (!h) return EB_ErrorInvalidComponent;
    EbErrorType e = EB_ErrorNone;
    if (h->p_component_private) {
        (void)eb_vp9_deinit_encoder(h);
        free(h->p_component_private);
    } else {
        e = EB_ErrorUndefined;
    }
    free(h);
    return e;
}

/* part of the reference library's exported surface (EB_API, Codec/EbEncHandle.c:3086-3110): its sample application links it for its
 * own string handling (App/EbAppConfig.c) */
size_t eb_vp9_strnlen_ss(const char *str, size_t max_len) {
    if (!str || max_len == 0 || max_len > (4ul << 10)) return 0;
    size_t n = 0;
    while (n < max_len && str[n]) n++;
    return n;
}

/* ---- extensions ---- */
EbErrorType svt_vp9_shim_set_mode_decision(EbComponentType *h, svt_vp9_shim_md_callback cb, void *user) {
    shim_state *s = state_of(h);
    if (!s || s->initialised) return EB_ErrorBadParameter; /* before eb_vp9_init_encoder */
    s->md_cb = cb;
    s->md_user = user;
    return EB_ErrorNone;
}

EbErrorType svt_vp9_shim_get_me_results(EbComponentType *h, uint64_t picture_number, svt_vp9_shim_picture_info *info, void *out,
                                        uint64_t out_bytes) {
    shim_state *s = state_of(h);
    if (!s || !s->initialised) return EB_ErrorBadParameter;
    { const EbErrorType je = join_all(s); if (je != EB_ErrorNone) return je; }
    shim_dev  *d = NULL;
    shim_slot *t = find_any(s, (int64_t)picture_number, &d);
    /* a picture that waits in an incomplete mini-GOP has no results yet; neither has one the ring has already given away */
    if (!t || !t->processed) return EB_NoErrorEmptyQueue;
    GPU_TRY(svt_hip_ctx_marker_wait(t->marker_ctx ? t->marker_ctx : d->ctx, t->marker));
    if (info) *info = t->info;
    if (out && !t->info.is_intra) {
        const uint64_t need = (uint64_t)t->info.n_sb * 85 * sizeof(svt_me_pu_result);
        if (out_bytes < need) return EB_ErrorBadParameter;
        GPU_TRY(svt_hip_mem_download(d->ctx, out, t->d_results, (size_t)need));
    }
    return EB_ErrorNone;
}

EbErrorType svt_vp9_shim_get_sb_stats(EbComponentType *h, uint64_t picture_number, void *stats, uint64_t stats_bytes, uint32_t *histograms,
                                      uint8_t *mean, uint16_t *variance) {
    shim_state *s = state_of(h);
    if (!s || !s->initialised) return EB_ErrorBadParameter;
    { const EbErrorType je = join_all(s); if (je != EB_ErrorNone) return je; }
    shim_dev  *d = NULL;
    shim_slot *t = find_any(s, (int64_t)picture_number, &d);
    if (!t || !t->processed) return EB_NoErrorEmptyQueue;
    GPU_TRY(svt_hip_ctx_marker_wait(t->marker_ctx ? t->marker_ctx : d->ctx, t->marker));
    const size_t n_sb = t->info.n_sb;
    if (stats && !t->info.is_intra) {
        if (stats_bytes < n_sb * sizeof(svt_me_sb_stats)) return EB_ErrorBadParameter;
        GPU_TRY(svt_hip_mem_download(d->ctx, stats, t->d_stats, n_sb * sizeof(svt_me_sb_stats)));
    }
    if (histograms && !t->info.is_intra) GPU_TRY(svt_hip_mem_download(d->ctx, histograms, t->d_hist, (2 * SVT_SAD_INTERVALS + 1) * sizeof(uint32_t)));
    if (mean) GPU_TRY(svt_hip_mem_download(d->ctx, mean, t->d_mean, n_sb * 85));
    if (variance) GPU_TRY(svt_hip_mem_download(d->ctx, variance, t->d_var, n_sb * 85 * sizeof(uint16_t)));
    return EB_ErrorNone;
}

EbErrorType svt_vp9_shim_get_coded_picture(EbComponentType *h, uint64_t picture_number, svt_vp9_shim_picture_info *info, void *mc_mode_info,
                                           void *lf_mode_info, int16_t *qcoeff, uint16_t *eob_map) {
    shim_state *s = state_of(h);
    if (!s || !s->initialised) return EB_ErrorBadParameter;
    { const EbErrorType je = join_all(s); if (je != EB_ErrorNone) return je; }
    shim_dev  *d = NULL;
    shim_slot *t = find_any(s, (int64_t)picture_number, &d);
    if (!t || !t->coded) return EB_NoErrorEmptyQueue;
    GPU_TRY(svt_hip_ctx_marker_wait(t->marker_ctx ? t->marker_ctx : d->ctx, t->marker));
    if (info) *info = t->info;
    const size_t units = (size_t)s->mi_rows * s->mi_cols;
    if (mc_mode_info) GPU_TRY(svt_hip_mem_download(d->ctx, mc_mode_info, t->d_mc_mi, units * sizeof(svt_mc_mode_info)));
    if (lf_mode_info) GPU_TRY(svt_hip_mem_download(d->ctx, lf_mode_info, t->d_lf_mi, units * sizeof(svt_lf_mode_info)));
    if (qcoeff) GPU_TRY(svt_hip_mem_download(d->ctx, qcoeff, t->d_qcoeff, s->coeffs * sizeof(int16_t)));
    if (eob_map) GPU_TRY(svt_hip_mem_download(d->ctx, eob_map, t->d_eob_map, (size_t)(s->W / 4) * (s->H / 4) * 3 / 2 * sizeof(uint16_t)));
    return EB_ErrorNone;
}

EbErrorType svt_vp9_shim_get_reference_picture(EbComponentType *h, uint64_t picture_number, uint8_t *out, uint64_t bytes) {
    shim_state *s = state_of(h);
    if (!s || !s->initialised || !out) return EB_ErrorBadParameter;
    { const EbErrorType je = join_all(s); if (je != EB_ErrorNone) return je; }
    shim_dev  *d = NULL;
    shim_slot *t = find_any(s, (int64_t)picture_number, &d);
    if (!t || !t->coded) return EB_NoErrorEmptyQueue;
    const size_t pw = (size_t)s->W + 2 * SHIM_REF_PAD, ph = (size_t)s->H + 2 * SHIM_REF_PAD, cpw = (size_t)s->W / 2 + SHIM_REF_PAD, cph = (size_t)s->H / 2 + SHIM_REF_PAD;
    const size_t need = pw * ph + 2 * cpw * cph;
    if (bytes < need) return EB_ErrorBadParameter;
    GPU_TRY(svt_hip_ctx_marker_wait(t->marker_ctx ? t->marker_ctx : d->ctx, t->marker));
    GPU_TRY(svt_hip_mem_download(d->ctx, out, t->d_rec, need));
    return EB_ErrorNone;
}

EbErrorType svt_vp9_shim_get_counters(EbComponentType *h, uint64_t *me_launches, uint64_t *pictures_sent) {
    shim_state *s = state_of(h);
    if (!s) return EB_ErrorBadParameter;
    (void)join_all(s);
    if (me_launches) *me_launches = s->me_launches;
    if (pictures_sent) *pictures_sent = (uint64_t)s->next_number;
    return EB_ErrorNone;
}
