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
 *   eb_vp9_svt_get_recon         EB_ErrorMax when recon_file == 0 (:2856-2861)
 *   stream_header / eos_nal      no-ops returning EB_ErrorNone (:2953-2971)
 * What the library does with the pictures: picture analysis (padded / decimated planes, block mean / variance) as each picture
 * arrives, motion estimation of a whole mini-GOP in ONE batched launch (svt_hip_me_batch_layers_device) plus the per-SB ME
 * statistics (svt_hip_me_sb_stats_device) when its last picture has arrived -- all enqueued asynchronously: send_picture returns
 * after the host copy of the picture (pinned staging), get_packet polls completion markers and blocks only when the caller says
 * it has sent its last picture (pic_send_done), as in the reference (:2880-2915).  Every picture is answered by a zero-byte
 * packet: entropy coding is outside the hot path.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/svt_vp9_enc_api.h"
#include "../../include/svtvp9_hip.h"

#define SHIM_MAX_MINIGOP 16

typedef struct shim_packet {
    EbBufferHeaderType  hdr;
    struct shim_packet *next;
    uint64_t            marker;     /* the GPU work behind this packet (svt_hip_ctx_marker_*) */
} shim_packet;

typedef struct shim_slot { /* one buffered picture: its three ME planes, its block statistics and its ME outputs, all on the device */
    int64_t        number;  /* display order, -1 = empty */
    int64_t        pts;
    svt_pa_picture pa;
    void          *d_luma;  /* the source luma, tightly packed (stride = width) */
    void          *d_results, *d_mean, *d_var, *d_rcme, *d_stats, *d_hist;
    int            processed;   /* its ME (or, for an intra picture, its analysis) has been enqueued */
    int            has_marker;
    uint64_t       marker;      /* completion of everything enqueued for this picture so far */
    svt_vp9_shim_picture_info info;
} shim_slot;

typedef struct shim_state {
    EbSvtVp9EncConfiguration cfg;   /* the library's copy, with frame_rate / intra_period resolved (copy_api_from_app) */
    int         configured, initialised, eos;
    int         levels, minigop;       /* hierarchical levels, 1 << levels */
    int         intra_period;          /* resolved */
    svt_hip_ctx *ctx;
    int         n_slots;
    shim_slot   *slot;
    int64_t     next_number;           /* display number of the next picture sent */
    int64_t     pending_first;         /* first picture of the mini-GOP being collected */
    int         pending;               /* pictures collected */
    int64_t     last_base;             /* display number of the latest base-layer / intra picture (-1: none yet) */
    shim_packet *q_head, *q_tail;
    int         eos_reported;
    uint64_t    me_launches;           /* batched ME launches so far (svt_vp9_shim_get_counters) */
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

static shim_state *state_of(EbComponentType *h) { return h ? (shim_state *)h->p_component_private : NULL; }

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

static void free_slots(shim_state *s) {
    if (!s->slot) return;
    for (int i = 0; i < s->n_slots; i++) {
        shim_slot *t = &s->slot[i];
        svt_hip_mem_free(s->ctx, (void *)t->pa.full.buf);
        svt_hip_mem_free(s->ctx, (void *)t->pa.quarter.buf);
        svt_hip_mem_free(s->ctx, (void *)t->pa.sixteenth.buf);
        void *v[7] = {t->d_luma, t->d_results, t->d_mean, t->d_var, t->d_rcme, t->d_stats, t->d_hist};
        for (int k = 0; k < 7; k++) svt_hip_mem_free(s->ctx, v[k]);
    }
    free(s->slot);
    s->slot = NULL;
}

EbErrorType eb_vp9_init_encoder(EbComponentType *h) {
    shim_state *s = state_of(h);
    if (!s || !s->configured) return EB_ErrorBadParameter;
    if (s->initialised) return EB_ErrorNone;
    /* target_socket names a CPU socket in the reference (-1 = both); here the GPU ordinal comes from SVT_HIP_DEVICE (default 0) */
    const char *dv = getenv("SVT_HIP_DEVICE");
    if (svt_hip_ctx_create(&s->ctx, dv ? atoi(dv) : 0) != SVT_HIP_OK) {
        fprintf(stderr, "SvtVp9Enc (GPU hot path): %s\n", svt_hip_last_error());
        s->ctx = NULL;
        return EB_ErrorInsufficientResources;
    }
    const int W = (int)s->cfg.source_width, H = (int)s->cfg.source_height;
    const int pad[3] = {68, 32, 16}; /* PA reference paddings, Codec/EbEncHandle.c:1003-1026 */
    /* the mini-GOP being collected, the one whose ME is in flight, and the base picture before it */
    s->n_slots = 2 * s->minigop + 2;
    s->slot = (shim_slot *)calloc((size_t)s->n_slots, sizeof(shim_slot));
    if (!s->slot) { svt_hip_ctx_destroy(s->ctx); s->ctx = NULL; return EB_ErrorInsufficientResources; }
    const uint32_t n_sb = (uint32_t)svt_hip_sb_count(W, H);
    int ok = 1;
    for (int i = 0; ok && i < s->n_slots; i++) {
        shim_slot *t = &s->slot[i];
        t->number = -1;
        svt_plane *pl[3] = {&t->pa.full, &t->pa.quarter, &t->pa.sixteenth};
        for (int k = 0; ok && k < 3; k++) {
            const int w = W >> k, hh = H >> k;
            void *d = NULL;
            ok = svt_hip_mem_alloc(s->ctx, (size_t)(w + 2 * pad[k]) * (size_t)(hh + 2 * pad[k]), &d) == SVT_HIP_OK;
            pl[k]->buf = (const uint8_t *)d; pl[k]->stride = w + 2 * pad[k]; pl[k]->origin_x = pl[k]->origin_y = pad[k];
            pl[k]->width = w; pl[k]->height = hh;
        }
        ok = ok && svt_hip_mem_alloc(s->ctx, (size_t)W * H, &t->d_luma) == SVT_HIP_OK &&
             svt_hip_mem_alloc(s->ctx, (size_t)n_sb * 85 * sizeof(svt_me_pu_result), &t->d_results) == SVT_HIP_OK &&
             svt_hip_mem_alloc(s->ctx, (size_t)n_sb * 85, &t->d_mean) == SVT_HIP_OK &&
             svt_hip_mem_alloc(s->ctx, (size_t)n_sb * 85 * sizeof(uint16_t), &t->d_var) == SVT_HIP_OK &&
             svt_hip_mem_alloc(s->ctx, (size_t)n_sb * sizeof(uint32_t), &t->d_rcme) == SVT_HIP_OK &&
             svt_hip_mem_alloc(s->ctx, (size_t)n_sb * sizeof(svt_me_sb_stats), &t->d_stats) == SVT_HIP_OK &&
             svt_hip_mem_alloc(s->ctx, (2 * SVT_SAD_INTERVALS + 1) * sizeof(uint32_t), &t->d_hist) == SVT_HIP_OK;
        t->info.n_sb = n_sb;
    }
    if (!ok) { /* nothing half-initialised is left behind: the handle is back in its configured state */
        free_slots(s);
        svt_hip_ctx_destroy(s->ctx);
        s->ctx = NULL;
        return EB_ErrorInsufficientResources;
    }
    {   /* take the context's pinned staging buffers now (init is outside the clock of SURVEY 8(d)'s metric, the first pictures are
           not): one throw-away upload per buffer of the ring */
        uint8_t *z = (uint8_t *)calloc((size_t)W, (size_t)H);
        if (z) {
            for (int i = 0; i < 4; i++) (void)svt_hip_mem_upload_2d_async(s->ctx, s->slot[0].d_luma, (size_t)W, z, (size_t)W, (size_t)W, (size_t)H);
            (void)svt_hip_ctx_synchronize(s->ctx);
            free(z);
        }
    }
    s->initialised = 1;
    return EB_ErrorNone;
}

EbErrorType eb_vp9_svt_enc_stream_header(EbComponentType *h, EbBufferHeaderType **o) { (void)h; (void)o; return EB_ErrorNone; }
EbErrorType eb_vp9_svt_enc_eos_nal(EbComponentType *h, EbBufferHeaderType **o) { (void)h; (void)o; return EB_ErrorNone; }

/* ------------------------------------------------------------------------------------------------ */
static shim_slot *slot_of(shim_state *s, int64_t number) { return &s->slot[number % s->n_slots]; }

static int push_packet(shim_state *s, int64_t pts, uint32_t flags, uint32_t pic_type, uint64_t marker) {
    shim_packet *p = (shim_packet *)calloc(1, sizeof *p);
    if (!p) return -1;
    p->hdr.size = sizeof(EbBufferHeaderType);
    p->hdr.pts = p->hdr.dts = pts;
    p->hdr.flags = flags;
    p->hdr.pic_type = pic_type;
    p->hdr.wrapper_ptr = p; /* the round trip of the reference's wrapper_ptr (:2923) */
    p->marker = marker;
    if (s->q_tail) s->q_tail->next = p; else s->q_head = p;
    s->q_tail = p;
    return 0;
}

/* one picture of a group whose motion estimation is about to be launched */
typedef struct shim_job {
    int64_t       number, ref0, ref1;
    int           layer, levels, n_lists, used_as_ref;
    svt_me_params p;
} shim_job;

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

/* do two parameter sets belong to one launch (svt_hip_me_batch_layers_device)?  They may differ in the four per-picture fields */
static int same_launch(const svt_me_params *a, const svt_me_params *b) {
    svt_me_params x = *a, y = *b;
    x.num_ref_lists = y.num_ref_lists = 0; x.temporal_layer_index = y.temporal_layer_index = 0;
    x.hierarchical_levels = y.hierarchical_levels = 0; x.same_ref_poc = y.same_ref_poc = 0;
    return memcmp(&x, &y, sizeof x) == 0;
}

/* the hierarchy between two already-listed pictures lo < hi by bisection, decode order */
static void add_hierarchy(shim_job *jobs, int *n, int64_t lo, int64_t hi, int layer, int levels) {
    if (hi - lo < 2) return;
    const int64_t mid = (lo + hi) / 2;
    shim_job *j = &jobs[(*n)++];
    memset(j, 0, sizeof *j);
    j->number = mid; j->ref0 = lo; j->ref1 = hi; j->layer = layer; j->levels = levels; j->n_lists = 2; j->used_as_ref = layer < levels;
    add_hierarchy(jobs, n, lo, mid, layer + 1, levels);
    add_hierarchy(jobs, n, mid, hi, layer + 1, levels);
}

/* Motion estimation of the collected group: its pictures are cut into parts as the reference cuts them
 * (svt_hip_minigop_split), all of them go to the GPU in as few launches as their parameter sets allow (one for a regular
 * mini-GOP), followed by the per-SB statistics of every picture.  Nothing here waits for the device. */
static EbErrorType flush_pending(shim_state *s, int cut_by_intra, int end_of_stream) {
    if (!s->pending) return EB_ErrorNone;
    shim_job jobs[SHIM_MAX_MINIGOP];
    int      n = 0;
    const int64_t first = s->pending_first;
    svt_minigop_part parts[4];
    const int np = svt_hip_minigop_split(s->pending, s->levels, cut_by_intra, parts);
    if (np < 1) return EB_ErrorBadParameter;
    int64_t prev = s->last_base;
    for (int k = 0; k < np; k++) {
        const int64_t p0 = first + parts[k].start, base = p0 + parts[k].length - 1;
        if (parts[k].random_access && prev >= 0) { /* base picture first (decode order), then the B hierarchy */
            shim_job *j = &jobs[n++];
            memset(j, 0, sizeof *j);
            j->number = base; j->ref0 = j->ref1 = prev; j->layer = 0; j->levels = parts[k].hierarchical_levels; j->n_lists = 2; j->used_as_ref = 1;
            add_hierarchy(jobs, &n, prev, base, 1, parts[k].hierarchical_levels);
        } else { /* low-delay P: the reference's structure tables (Codec/EbPredictionStructure.c) are picture decision, not
                    reproduced -- every picture is predicted from its predecessor and serves as the next one's reference */
            for (int64_t q = p0; q <= base; q++) {
                shim_job *j = &jobs[n++];
                memset(j, 0, sizeof *j);
                j->number = q; j->ref0 = q - 1; j->ref1 = -1; j->layer = 0; j->levels = parts[k].hierarchical_levels; j->n_lists = 1; j->used_as_ref = 1;
            }
        }
        prev = base;
    }
    for (int i = 0; i < n; i++) {
        const EbErrorType e = job_params(s, &jobs[i]);
        if (e != EB_ErrorNone) return e;
    }
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
            shim_slot *t = slot_of(s, jobs[k].number);
            cur[m] = t->pa; r0[m] = slot_of(s, jobs[k].ref0)->pa; r1[m] = slot_of(s, jobs[k].n_lists == 2 ? jobs[k].ref1 : jobs[k].ref0)->pa;
            pp[m] = jobs[k].p; res[m] = (svt_me_pu_result *)t->d_results; rc[m] = (uint32_t *)t->d_rcme;
            done[k] = 1;
            m++;
        }
        if (svt_hip_me_batch_layers_device(s->ctx, m, cur, r0, r1, pp, res, s->cfg.rate_control_mode ? rc : NULL) != SVT_HIP_OK) {
            fprintf(stderr, "SvtVp9Enc (GPU hot path): %s\n", svt_hip_last_error());
            return EB_ErrorMax;
        }
        s->me_launches++;
    }
    /* the tail of the ME kernel process per picture (Codec/EbMotionEstimationProcess.c:1047-1237): stationary-edge flags and the
       rate-control histograms from the ME results and the picture-analysis variances, all device resident */
    const int res_class = svt_hip_input_resolution((int32_t)s->cfg.source_width, (int32_t)s->cfg.source_height);
    for (int i = 0; i < n; i++) {
        shim_slot *t = slot_of(s, jobs[i].number);
        svt_me_sb_stats_params sp;
        memset(&sp, 0, sizeof sp);
        sp.pic_width = (int32_t)s->cfg.source_width; sp.pic_height = (int32_t)s->cfg.source_height; sp.input_resolution = res_class;
        sp.temporal_layer_index = jobs[i].layer; sp.slice_type = jobs[i].n_lists == 2 ? 0 : 1;
        sp.run_part2 = !end_of_stream; sp.rate_control_mode = (int32_t)s->cfg.rate_control_mode;
        if (svt_hip_mem_set(s->ctx, t->d_hist, 0, (2 * SVT_SAD_INTERVALS + 1) * sizeof(uint32_t)) != SVT_HIP_OK ||
            svt_hip_me_sb_stats_device(s->ctx, &sp, (const svt_me_pu_result *)t->d_results, (const uint16_t *)t->d_var,
                                       s->cfg.rate_control_mode ? (const uint32_t *)t->d_rcme : NULL, (svt_me_sb_stats *)t->d_stats, (uint32_t *)t->d_hist,
                                       (uint32_t *)t->d_hist + 2 * SVT_SAD_INTERVALS) != SVT_HIP_OK) {
            fprintf(stderr, "SvtVp9Enc (GPU hot path): %s\n", svt_hip_last_error());
            return EB_ErrorMax;
        }
    }
    uint64_t marker = 0;
    if (svt_hip_ctx_marker_record(s->ctx, &marker) != SVT_HIP_OK) return EB_ErrorMax;
    for (int i = 0; i < n; i++) { /* packets in decode order */
        shim_slot *t = slot_of(s, jobs[i].number);
        t->info.is_intra = 0; t->info.temporal_layer_index = jobs[i].layer; t->info.hierarchical_levels = jobs[i].levels;
        t->info.num_ref_lists = jobs[i].n_lists;
        t->info.ref_picture_number[0] = jobs[i].ref0; t->info.ref_picture_number[1] = jobs[i].n_lists == 2 ? jobs[i].ref1 : -1;
        t->processed = 1; t->has_marker = 1; t->marker = marker;
        if (push_packet(s, t->pts, 0, jobs[i].n_lists == 2 ? 0 /* EB_B_PICTURE */ : 1 /* EB_P_PICTURE */, marker)) return EB_ErrorInsufficientResources;
    }
    s->last_base = first + s->pending - 1;
    s->pending = 0;
    return EB_ErrorNone;
}

EbErrorType eb_vp9_svt_enc_send_picture(EbComponentType *h, EbBufferHeaderType *b) {
    shim_state *s = state_of(h);
    if (!s || !s->initialised) return EB_ErrorBadParameter;
    if (s->eos) return EB_ErrorBadParameter;
    const int end = !b || !b->p_buffer || (b->flags & EB_BUFFERFLAG_EOS);
    EbErrorType e = EB_ErrorNone;
    if (b && b->p_buffer) {
        const EbSvtEncInput *in = (const EbSvtEncInput *)b->p_buffer;
        if (!in->luma || in->y_stride < s->cfg.source_width) return EB_ErrorBadParameter;
        const int64_t n = s->next_number;
        shim_slot    *t = slot_of(s, n);
        const int     W = (int)s->cfg.source_width, H = (int)s->cfg.source_height;
        /* the slot's previous picture (2 mini-GOPs + 2 ago) must have left the GPU: the one place send_picture can block, as the
           reference blocks when its picture pool is exhausted */
        if (t->has_marker && svt_hip_ctx_marker_wait(s->ctx, t->marker) != SVT_HIP_OK) return EB_ErrorMax;
        /* the copy the reference makes in copy_frame_buffer (:2743-2796), into pinned staging: the caller's planes are free again
           on return; the transfer and the analysis below run asynchronously */
        if (svt_hip_mem_upload_2d_async(s->ctx, t->d_luma, (size_t)W, in->luma, in->y_stride, (size_t)W, (size_t)H) != SVT_HIP_OK) return EB_ErrorMax;
        const uint8_t *lum = (const uint8_t *)t->d_luma;
        const int32_t  stride = W;
        if (svt_hip_pa_prepare_batch_device(s->ctx, 1, &lum, &stride, &t->pa, 1) != SVT_HIP_OK ||
            svt_hip_pa_mean_variance_device(s->ctx, &t->pa.full, (uint8_t *)t->d_mean, (uint16_t *)t->d_var) != SVT_HIP_OK) return EB_ErrorMax;
        /* the picture is accepted from here on */
        s->next_number = n + 1;
        t->number = n; t->pts = b->pts; t->info.picture_number = (uint64_t)n; t->processed = 0; t->has_marker = 0;
        const int intra = n == 0 || (s->intra_period >= 0 && n % (s->intra_period + 1) == 0);
        if (intra) {
            if ((e = flush_pending(s, 1, 0)) != EB_ErrorNone) return e;
            t->info.is_intra = 1; t->info.num_ref_lists = 0; t->info.temporal_layer_index = 0; t->info.hierarchical_levels = s->levels;
            t->info.ref_picture_number[0] = t->info.ref_picture_number[1] = -1;
            if (svt_hip_ctx_marker_record(s->ctx, &t->marker) != SVT_HIP_OK) return EB_ErrorMax;
            t->has_marker = 1; t->processed = 1;
            if (push_packet(s, t->pts, 0, 2 /* EB_I_PICTURE */, t->marker)) return EB_ErrorInsufficientResources;
            s->last_base = n;
        } else {
            if (!s->pending) s->pending_first = n;
            if (++s->pending == s->minigop) e = flush_pending(s, 0, 0);
        }
    }
    if (end && e == EB_ErrorNone) {
        e = flush_pending(s, 0, 1);
        s->eos = 1;
        if (e == EB_ErrorNone) {
            if (s->q_tail) s->q_tail->hdr.flags |= EB_BUFFERFLAG_EOS; /* the last picture's packet closes the stream */
            else {
                uint64_t m = 0;
                if (svt_hip_ctx_marker_record(s->ctx, &m) != SVT_HIP_OK) return EB_ErrorMax;
                if (push_packet(s, b ? b->pts : 0, EB_BUFFERFLAG_EOS, 0, m)) e = EB_ErrorInsufficientResources;
            }
        }
    }
    return e;
}

/* Non-blocking while pictures are still being sent (EB_NoErrorEmptyQueue when the next packet's GPU work has not finished);
 * with pic_send_done the call waits for it, as eb_vp9_svt_get_packet does (:2880-2915: eb_vp9_get_full_object vs the
 * non-blocking variant). */
EbErrorType eb_vp9_svt_get_packet(EbComponentType *h, EbBufferHeaderType **p_buffer, uint8_t pic_send_done) {
    shim_state *s = state_of(h);
    if (!s || !p_buffer) return EB_ErrorBadParameter;
    shim_packet *p = s->q_head;
    if (!p) return EB_NoErrorEmptyQueue;
    if (s->ctx) {
        if (pic_send_done) { if (svt_hip_ctx_marker_wait(s->ctx, p->marker) != SVT_HIP_OK) return EB_ErrorMax; }
        else {
            const int32_t q = svt_hip_ctx_marker_query(s->ctx, p->marker);
            if (q < 0) return EB_ErrorMax;
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

EbErrorType eb_vp9_svt_get_recon(EbComponentType *h, EbBufferHeaderType *p_buffer) {
    shim_state *s = state_of(h);
    (void)p_buffer;
    if (!s) return EB_ErrorBadParameter;
    return s->cfg.recon_file ? EB_NoErrorEmptyQueue : EB_ErrorMax;
}

EbErrorType eb_vp9_deinit_encoder(EbComponentType *h) {
    shim_state *s = state_of(h);
    if (!s) return EB_ErrorNone; /* the reference accepts a NULL component here (:1846) */
    while (s->q_head) { shim_packet *p = s->q_head; s->q_head = p->next; free(p); }
    s->q_tail = NULL;
    if (s->ctx) {
        free_slots(s);
        svt_hip_ctx_destroy(s->ctx);
        s->ctx = NULL;
    }
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

EbErrorType svt_vp9_shim_get_me_results(EbComponentType *h, uint64_t picture_number, svt_vp9_shim_picture_info *info, void *out,
                                        uint64_t out_bytes) {
    shim_state *s = state_of(h);
    if (!s || !s->initialised) return EB_ErrorBadParameter;
    shim_slot *t = slot_of(s, (int64_t)picture_number);
    /* a picture that waits in an incomplete mini-GOP has no results yet; neither has one the ring has already given away */
    if (t->number != (int64_t)picture_number || !t->processed) return EB_NoErrorEmptyQueue;
    if (svt_hip_ctx_marker_wait(s->ctx, t->marker) != SVT_HIP_OK) return EB_ErrorMax;
    if (info) *info = t->info;
    if (out && !t->info.is_intra) {
        const uint64_t need = (uint64_t)t->info.n_sb * 85 * sizeof(svt_me_pu_result);
        if (out_bytes < need) return EB_ErrorBadParameter;
        if (svt_hip_mem_download(s->ctx, out, t->d_results, (size_t)need) != SVT_HIP_OK) return EB_ErrorMax;
    }
    return EB_ErrorNone;
}

EbErrorType svt_vp9_shim_get_sb_stats(EbComponentType *h, uint64_t picture_number, void *stats, uint64_t stats_bytes, uint32_t *histograms,
                                      uint8_t *mean, uint16_t *variance) {
    shim_state *s = state_of(h);
    if (!s || !s->initialised) return EB_ErrorBadParameter;
    shim_slot *t = slot_of(s, (int64_t)picture_number);
    if (t->number != (int64_t)picture_number || !t->processed) return EB_NoErrorEmptyQueue;
    if (svt_hip_ctx_marker_wait(s->ctx, t->marker) != SVT_HIP_OK) return EB_ErrorMax;
    const size_t n_sb = t->info.n_sb;
    if (stats && !t->info.is_intra) {
        if (stats_bytes < n_sb * sizeof(svt_me_sb_stats)) return EB_ErrorBadParameter;
        if (svt_hip_mem_download(s->ctx, stats, t->d_stats, n_sb * sizeof(svt_me_sb_stats)) != SVT_HIP_OK) return EB_ErrorMax;
    }
    if (histograms && !t->info.is_intra && svt_hip_mem_download(s->ctx, histograms, t->d_hist, (2 * SVT_SAD_INTERVALS + 1) * sizeof(uint32_t)) != SVT_HIP_OK) return EB_ErrorMax;
    if (mean && svt_hip_mem_download(s->ctx, mean, t->d_mean, n_sb * 85) != SVT_HIP_OK) return EB_ErrorMax;
    if (variance && svt_hip_mem_download(s->ctx, variance, t->d_var, n_sb * 85 * sizeof(uint16_t)) != SVT_HIP_OK) return EB_ErrorMax;
    return EB_ErrorNone;
}

EbErrorType svt_vp9_shim_get_counters(EbComponentType *h, uint64_t *me_launches, uint64_t *pictures_sent) {
    shim_state *s = state_of(h);
    if (!s) return EB_ErrorBadParameter;
    if (me_launches) *me_launches = s->me_launches;
    if (pictures_sent) *pictures_sent = (uint64_t)s->next_number;
    return EB_ErrorNone;
}
