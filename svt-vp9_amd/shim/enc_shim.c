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
 * What the library does with the pictures: picture analysis + motion estimation of the reference's random-access
 * mini-GOPs on the GPU (see svt_vp9_enc_api.h); every picture is answered by a zero-byte packet.
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
} shim_packet;

typedef struct shim_slot { /* one buffered picture: its three ME planes and its ME results, all on the device */
    int64_t        number;  /* display order, -1 = empty */
    int64_t        pts;
    svt_pa_picture pa;
    void          *d_luma;  /* the source luma, tightly packed (stride = width) */
    void          *d_results;
    svt_vp9_shim_picture_info info;
} shim_slot;

typedef struct shim_state {
    EbSvtVp9EncConfiguration cfg;
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
    if (verify(p) != EB_ErrorNone) return EB_ErrorBadParameter;
    s->cfg = *p;
    /* set_param_based_on_input (:2166-2192) */
    s->levels  = (s->cfg.tune != 0 && s->cfg.rate_control_mode == 0) ? 4 : 3;
    s->minigop = 1 << s->levels;
    if (s->cfg.frame_rate_numerator != 0 && s->cfg.frame_rate_denominator != 0)
        s->cfg.frame_rate = ((s->cfg.frame_rate_numerator << 8) / s->cfg.frame_rate_denominator) << 8;
    s->intra_period = s->cfg.intra_period;
    if (s->intra_period == -2) { /* compute_default_intra_period (:2014-2024) */
        const int fps = s->cfg.frame_rate < 1000 ? (int)s->cfg.frame_rate : (int)(s->cfg.frame_rate >> 16);
        const int lo = fps / s->minigop * s->minigop, hi = (fps + s->minigop) / s->minigop * s->minigop;
        s->intra_period = abs(fps - hi) > abs(fps - lo) ? lo : hi;
    }
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
        svt_hip_mem_free(s->ctx, t->d_luma);
        svt_hip_mem_free(s->ctx, t->d_results);
    }
    free(s->slot);
    s->slot = NULL;
}

EbErrorType eb_vp9_init_encoder(EbComponentType *h) {
    shim_state *s = state_of(h);
    if (!s || !s->configured) return EB_ErrorBadParameter;
    if (s->initialised) return EB_ErrorNone;
    if (svt_hip_ctx_create(&s->ctx, s->cfg.target_socket > 0 ? s->cfg.target_socket : 0) != SVT_HIP_OK) {
        fprintf(stderr, "SvtVp9Enc (GPU hot path): %s\n", svt_hip_last_error());
        s->ctx = NULL;
        return EB_ErrorInsufficientResources;
    }
    const int W = (int)s->cfg.source_width, H = (int)s->cfg.source_height;
    const int pad[3] = {68, 32, 16}; /* PA reference paddings, Codec/EbEncHandle.c:1003-1026 */
    s->n_slots = s->minigop + 1;
    s->slot = (shim_slot *)calloc((size_t)s->n_slots, sizeof(shim_slot));
    if (!s->slot) return EB_ErrorInsufficientResources;
    const uint32_t n_sb = (uint32_t)svt_hip_sb_count(W, H);
    for (int i = 0; i < s->n_slots; i++) {
        shim_slot *t = &s->slot[i];
        t->number = -1;
        svt_plane *pl[3] = {&t->pa.full, &t->pa.quarter, &t->pa.sixteenth};
        for (int k = 0; k < 3; k++) {
            const int w = W >> k, hh = H >> k;
            void *d = NULL;
            if (svt_hip_mem_alloc(s->ctx, (size_t)(w + 2 * pad[k]) * (size_t)(hh + 2 * pad[k]), &d) != SVT_HIP_OK) { free_slots(s); return EB_ErrorInsufficientResources; }
            pl[k]->buf = (const uint8_t *)d; pl[k]->stride = w + 2 * pad[k]; pl[k]->origin_x = pl[k]->origin_y = pad[k];
            pl[k]->width = w; pl[k]->height = hh;
        }
        if (svt_hip_mem_alloc(s->ctx, (size_t)W * H, &t->d_luma) != SVT_HIP_OK ||
            svt_hip_mem_alloc(s->ctx, (size_t)n_sb * 85 * sizeof(svt_me_pu_result), &t->d_results) != SVT_HIP_OK) { free_slots(s); return EB_ErrorInsufficientResources; }
        t->info.n_sb = n_sb;
    }
    s->initialised = 1;
    return EB_ErrorNone;
}

EbErrorType eb_vp9_svt_enc_stream_header(EbComponentType *h, EbBufferHeaderType **o) { (void)h; (void)o; return EB_ErrorNone; }
EbErrorType eb_vp9_svt_enc_eos_nal(EbComponentType *h, EbBufferHeaderType **o) { (void)h; (void)o; return EB_ErrorNone; }

/* ------------------------------------------------------------------------------------------------ */
static shim_slot *slot_of(shim_state *s, int64_t number) { return &s->slot[number % s->n_slots]; }

static int push_packet(shim_state *s, int64_t pts, uint32_t flags, uint32_t pic_type) {
    shim_packet *p = (shim_packet *)calloc(1, sizeof *p);
    if (!p) return -1;
    p->hdr.size = sizeof(EbBufferHeaderType);
    p->hdr.pts = p->hdr.dts = pts;
    p->hdr.flags = flags;
    p->hdr.pic_type = pic_type;
    p->hdr.wrapper_ptr = p; /* the round trip of the reference's wrapper_ptr (:2923) */
    if (s->q_tail) s->q_tail->next = p; else s->q_head = p;
    s->q_tail = p;
    return 0;
}

/* motion estimation of one picture against its references, parameters as the reference derives them for this picture */
static EbErrorType me_picture(shim_state *s, int64_t number, int64_t ref0, int64_t ref1, int layer, int levels, int n_lists) {
    shim_slot *t = slot_of(s, number);
    svt_me_picture_config pc;
    memset(&pc, 0, sizeof pc);
    pc.pic_width = (int32_t)s->cfg.source_width; pc.pic_height = (int32_t)s->cfg.source_height;
    pc.enc_mode = s->cfg.enc_mode; pc.tune = s->cfg.tune;
    pc.frame_rate = (int32_t)(s->cfg.frame_rate > 1000 ? s->cfg.frame_rate >> 16 : s->cfg.frame_rate);
    pc.num_ref_lists = n_lists; pc.temporal_layer_index = layer; pc.hierarchical_levels = levels;
    pc.is_used_as_reference = layer < levels;
    pc.same_ref_poc = n_lists == 2 && ref0 == ref1;
    pc.rate_control_mode = (int32_t)s->cfg.rate_control_mode;
    svt_me_params p;
    if (svt_hip_me_params_derive(&p, &pc) != SVT_HIP_OK) return EB_ErrorBadParameter;
    if (!s->cfg.use_default_me_hme) { /* eb_vp9_set_me_hme_params_from_confi (Codec/EbMotionEstimationProcess.c:316-324) */
        p.search_area_width  = (uint8_t)s->cfg.search_area_width;
        p.search_area_height = (uint8_t)s->cfg.search_area_height;
        p.enable_hme_flag    = s->cfg.enable_hme_flag;
    }
    const svt_pa_picture *r1 = n_lists == 2 ? &slot_of(s, ref1)->pa : NULL;
    if (svt_hip_me_picture_device(s->ctx, &t->pa, &slot_of(s, ref0)->pa, r1, &p, (svt_me_pu_result *)t->d_results, NULL) != SVT_HIP_OK) {
        fprintf(stderr, "SvtVp9Enc (GPU hot path): %s\n", svt_hip_last_error());
        return EB_ErrorMax;
    }
    t->info.is_intra = 0; t->info.temporal_layer_index = layer; t->info.hierarchical_levels = levels; t->info.num_ref_lists = n_lists;
    t->info.ref_picture_number[0] = ref0; t->info.ref_picture_number[1] = n_lists == 2 ? ref1 : -1;
    return EB_ErrorNone;
}

/* a complete mini-GOP [first, first + n): base picture first (decode order), then the hierarchy by bisection */
static EbErrorType me_hierarchy(shim_state *s, int64_t lo, int64_t hi, int layer) { /* pictures strictly between lo and hi */
    if (hi - lo < 2) return EB_ErrorNone;
    const int64_t mid = (lo + hi) / 2;
    EbErrorType e = me_picture(s, mid, lo, hi, layer, s->levels, 2);
    if (e != EB_ErrorNone) return e;
    if (push_packet(s, slot_of(s, mid)->pts, 0, 0 /* EB_B_PICTURE */)) return EB_ErrorInsufficientResources;
    if ((e = me_hierarchy(s, lo, mid, layer + 1)) != EB_ErrorNone) return e;
    return me_hierarchy(s, mid, hi, layer + 1);
}

static EbErrorType flush_pending(shim_state *s) {
    EbErrorType e = EB_ErrorNone;
    if (!s->pending) return e;
    const int64_t first = s->pending_first;
    if (s->pending == s->minigop && s->last_base >= 0) {
        const int64_t base = first + s->minigop - 1;
        if ((e = me_picture(s, base, s->last_base, s->last_base, 0, s->levels, 2)) != EB_ErrorNone) return e;
        if (push_packet(s, slot_of(s, base)->pts, 0, 0)) return EB_ErrorInsufficientResources;
        if ((e = me_hierarchy(s, s->last_base, base, 1)) != EB_ErrorNone) return e;
        s->last_base = base;
    } else { /* a short group (end of stream, or cut by an intra refresh): a chain of P pictures, each from its predecessor */
        for (int k = 0; k < s->pending; k++) {
            if ((e = me_picture(s, first + k, first + k - 1, -1, 0, 0, 1)) != EB_ErrorNone) return e;
            if (push_packet(s, slot_of(s, first + k)->pts, 0, 1 /* EB_P_PICTURE */)) return EB_ErrorInsufficientResources;
        }
        s->last_base = first + s->pending - 1;
    }
    s->pending = 0;
    return e;
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
        const int64_t n = s->next_number++;
        shim_slot    *t = slot_of(s, n);
        const int     W = (int)s->cfg.source_width, H = (int)s->cfg.source_height;
        t->number = n; t->pts = b->pts; t->info.picture_number = (uint64_t)n;
        /* the copy the reference makes in copy_frame_buffer (:2743-2796): the caller's planes are free again on return */
        if (svt_hip_mem_upload_2d(s->ctx, t->d_luma, (size_t)W, in->luma, in->y_stride, (size_t)W, (size_t)H) != SVT_HIP_OK) return EB_ErrorMax;
        const uint8_t *lum = (const uint8_t *)t->d_luma;
        const int32_t  stride = W;
        if (svt_hip_pa_prepare_batch_device(s->ctx, 1, &lum, &stride, &t->pa, 1) != SVT_HIP_OK) return EB_ErrorMax;
        const int intra = n == 0 || (s->intra_period >= 0 && n % (s->intra_period + 1) == 0);
        if (intra) {
            if ((e = flush_pending(s)) != EB_ErrorNone) return e;
            t->info.is_intra = 1; t->info.num_ref_lists = 0; t->info.temporal_layer_index = 0; t->info.hierarchical_levels = s->levels;
            t->info.ref_picture_number[0] = t->info.ref_picture_number[1] = -1;
            if (push_packet(s, t->pts, 0, 2 /* EB_I_PICTURE */)) return EB_ErrorInsufficientResources;
            s->last_base = n;
        } else {
            if (!s->pending) s->pending_first = n;
            if (++s->pending == s->minigop) e = flush_pending(s);
        }
    }
    if (end && e == EB_ErrorNone) {
        e = flush_pending(s);
        s->eos = 1;
        if (e == EB_ErrorNone) {
            if (s->q_tail) s->q_tail->hdr.flags |= EB_BUFFERFLAG_EOS; /* the last picture's packet closes the stream */
            else if (push_packet(s, b ? b->pts : 0, EB_BUFFERFLAG_EOS, 0)) e = EB_ErrorInsufficientResources;
        }
    }
    if (e == EB_ErrorNone && svt_hip_ctx_synchronize(s->ctx) != SVT_HIP_OK) e = EB_ErrorMax;
    return e;
}

EbErrorType eb_vp9_svt_get_packet(EbComponentType *h, EbBufferHeaderType **p_buffer, uint8_t pic_send_done) {
    shim_state *s = state_of(h);
    (void)pic_send_done; /* everything sent has been processed when send_picture returned: nothing to block on */
    if (!s || !p_buffer) return EB_ErrorBadParameter;
    shim_packet *p = s->q_head;
    if (!p) return EB_NoErrorEmptyQueue;
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
    if (t->number != (int64_t)picture_number) return EB_NoErrorEmptyQueue;
    if (info) *info = t->info;
    if (out && !t->info.is_intra) {
        const uint64_t need = (uint64_t)t->info.n_sb * 85 * sizeof(svt_me_pu_result);
        if (out_bytes < need) return EB_ErrorBadParameter;
        if (svt_hip_mem_download(s->ctx, out, t->d_results, (size_t)need) != SVT_HIP_OK) return EB_ErrorMax;
    }
    return EB_ErrorNone;
}
