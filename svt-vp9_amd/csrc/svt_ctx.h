/* svt_ctx.h -- context object behind the C ABI (include/svtvp9_hip.h) and small launch helpers. */
#ifndef SVT_CTX_H
#define SVT_CTX_H
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include "../../include/svtvp9_hip.h"

#define SVT_CTX_SLOTS 40
#define SVT_CTX_RING 64
#define SVT_CTX_UPLOAD_RING 4   /* pinned staging buffers of svt_hip_mem_upload_2d_async */
#define SVT_CTX_MARKERS 1024    /* completion markers in flight (svt_hip_ctx_marker_*) */

struct svt_hip_ctx {
    int         device;
    hipStream_t stream;
    int         owns_stream;
    hipEvent_t  ev_start, ev_stop;
    int         timed;
    int         cu_count;  /* compute units of the device (queried once at creation) */
    int         me_instance; /* kernel instance of the last ME launch: me_spec.h index, + 100 for me_fast.h's driver, + 200 for the two launches of the compact layout (me_layout.h) */
    int         intra_wgs;   /* svt_hip_ctx_set_intra_workgroups: workgroups of the intra pass launched on this context (0 = default) */
    /* ring of staging slots (pinned host + device twin) for launch descriptors: a slot is reused only after the
       event recorded behind its last consumer has completed, so launches never synchronise the stream */
    void       *ring_host[SVT_CTX_RING];
    void       *ring_dev[SVT_CTX_RING];
    size_t      ring_bytes[SVT_CTX_RING];
    hipEvent_t  ring_ev[SVT_CTX_RING];
    int         ring_used[SVT_CTX_RING];
    int         ring_own[SVT_CTX_RING];   /* 1: the entry outgrew its share of the slabs below and owns its buffers */
    void       *ring_slab_host, *ring_slab_dev; /* SVT_CTX_RING x 64 KB, taken when the context is created (an entry used to take its buffers the
                                                   first time round the ring: 64 pairs of allocations inside the first pictures of a stream) */
    int         ring_pos;
    /* helper streams for entry points whose kernels are independent of each other (the four transform sizes of a TQ batch):
       forked from / joined into `stream` with events, so the call still behaves as one operation on the context's stream */
    hipStream_t aux[3];
    hipEvent_t  aux_fork, aux_join[3];
    int         aux_ready;
    /* asynchronous uploads: the caller's rows are copied into a pinned buffer of this ring before the call returns, the device
       copy is ordered on the stream; a buffer is reused once the event behind its copy has completed */
    void       *up_host[SVT_CTX_UPLOAD_RING];
    size_t      up_bytes[SVT_CTX_UPLOAD_RING];
    hipEvent_t  up_ev[SVT_CTX_UPLOAD_RING];
    int         up_used[SVT_CTX_UPLOAD_RING];
    int         up_pos;
    double      up_prof[3];     /* SVT_HIP_SHIM_PROFILE: seconds waiting for a staging slot / copying rows / enqueuing, and */
    long        up_prof_n;      /* the number of uploads (svt_hip_mem_upload_planes_async) */
    hipEvent_t  direct_ev;      /* behind the last upload that reads the caller's (page-locked) memory directly */
    int         direct_pending;
    /* completion markers: marker m is event m % SVT_CTX_MARKERS; before an event is recorded again its previous use is waited
       for, so a marker older than SVT_CTX_MARKERS records is complete by construction */
    hipEvent_t  mk_ev[SVT_CTX_MARKERS];
    uint64_t    mk_next;
    /* dedicated events of svt_hip_ref_handoff_device (producer side / consumer side) */
    hipEvent_t  ho_produced, ho_consumed;
    void       *slot[SVT_CTX_SLOTS]; /* grow-only device buffers of the host-pointer convenience entry points */
    size_t      slot_bytes[SVT_CTX_SLOTS];
};

int32_t svt_set_error(int32_t code, const char *msg);
int32_t svt_set_hip_error(hipError_t e, const char *file, int line);
/* next staging slot with at least `bytes` in both twins; returns 0 on success */
int     svt_ctx_stage(svt_hip_ctx *ctx, size_t bytes, void **host, void **dev);
/* call after the last operation that reads the slot has been enqueued on ctx->stream */
void    svt_ctx_stage_commit(svt_hip_ctx *ctx);
void   *svt_ctx_slot(svt_hip_ctx *ctx, int slot, size_t bytes);
/* creates the helper streams on first use (same priority as the context's stream); returns 0 on success */
int     svt_ctx_aux_init(svt_hip_ctx *ctx);

/* transform stage over device-built block lists (tq_kernel.hip; used by encdec.hip) */
int32_t svt_tq_launch_device_lists(svt_hip_ctx *ctx, const uint8_t *d_src, const uint8_t *d_pred, uint8_t *const *recon_set, int n_set,
                                   const svt_tq_block *d_blocks, const int32_t cap[4], const int32_t *d_off_cnt, const svt_quant_tables *d_qtabs,
                                   const int16_t *d_iscan, int16_t *d_qcoeff, int16_t *d_dqcoeff, uint16_t *d_eob, uint64_t *d_dist,
                                   const uint32_t *d_pos, const void *d_geom, int geom_stride, const uint32_t *d_iscan_off, int sb_cols);

/* the same over SB-ordered lists: one launch, workgroup = (picture, chunk of SVT_TQ_CHUNK_SBS SBs, size, part) (tq_kernel.hip: svt_tq_sb_kernel) */
#define SVT_TQ_CHUNK_SBS 4
int32_t svt_tq_launch_sb_lists(svt_hip_ctx *ctx, const uint8_t *d_src, const uint8_t *d_pred, uint8_t *const *recon_set, int n_set, const svt_quant_tables *d_qtabs,
                               const int16_t *d_iscan, int16_t *d_qcoeff, int16_t *d_dqcoeff, uint16_t *d_eob, const uint32_t *d_pos, const void *d_geom, int geom_stride,
                               const uint32_t *d_iscan_off, int sb_cols, const int32_t *d_seg, const int32_t *d_bases, int n_pics, int n_chunks, int seg_per_chunk);

/* encode pass of an intra picture / the stand-in intra decision (intra_kernel.hip; used by encdec.hip).  d_sync: 2 + 3 * (number of
 * 16x16 luma cells) dwords of scratch */
int32_t svt_intra_launch(svt_hip_ctx *ctx, const svt_encdec_picture *p, int32_t width, int32_t height, int32_t mi_stride, const svt_quant_tables *d_qtabs,
                         const int16_t *d_iscan, const uint32_t iscan_off[16], int32_t *d_sync, int32_t *d_status, int32_t mixed);
int32_t svt_md_intra_default_launch(svt_hip_ctx *ctx, svt_lf_mode_info *d_lf_mi, int32_t mi_stride, int32_t mi_rows, int32_t mi_cols, int32_t filter_level);

#define HIP_TRY(expr)                                                        \
    do {                                                                     \
        hipError_t e_ = (expr);                                              \
        if (e_ != hipSuccess) return svt_set_hip_error(e_, __FILE__, __LINE__); \
    } while (0)
#endif
