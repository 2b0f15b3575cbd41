/* svt_ctx.h -- context object behind the C ABI (include/svtvp9_hip.h) and small launch helpers. */
#ifndef SVT_CTX_H
#define SVT_CTX_H
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include "../../include/svtvp9_hip.h"

#define SVT_CTX_SLOTS 40
#define SVT_CTX_RING 64

struct svt_hip_ctx {
    int         device;
    hipStream_t stream;
    int         owns_stream;
    hipEvent_t  ev_start, ev_stop;
    int         timed;
    int         cu_count;  /* compute units of the device (queried once at creation) */
    /* ring of staging slots (pinned host + device twin) for launch descriptors: a slot is reused only after the
       event recorded behind its last consumer has completed, so launches never synchronise the stream */
    void       *ring_host[SVT_CTX_RING];
    void       *ring_dev[SVT_CTX_RING];
    size_t      ring_bytes[SVT_CTX_RING];
    hipEvent_t  ring_ev[SVT_CTX_RING];
    int         ring_used[SVT_CTX_RING];
    int         ring_pos;
    /* helper streams for entry points whose kernels are independent of each other (the four transform sizes of a TQ batch):
       forked from / joined into `stream` with events, so the call still behaves as one operation on the context's stream */
    hipStream_t aux[3];
    hipEvent_t  aux_fork, aux_join[3];
    int         aux_ready;
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

#define HIP_TRY(expr)                                                        \
    do {                                                                     \
        hipError_t e_ = (expr);                                              \
        if (e_ != hipSuccess) return svt_set_hip_error(e_, __FILE__, __LINE__); \
    } while (0)
#endif
