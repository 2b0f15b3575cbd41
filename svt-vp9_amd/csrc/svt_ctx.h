/* svt_ctx.h -- context object behind the C ABI (include/svtvp9_hip.h) and small launch helpers. */
#ifndef SVT_CTX_H
#define SVT_CTX_H
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include "../../include/svtvp9_hip.h"

#define SVT_CTX_SLOTS 24

struct svt_hip_ctx {
    int         device;
    hipStream_t stream;
    int         owns_stream;
    hipEvent_t  ev_start, ev_stop;
    int         timed;
    void       *host_scratch;  /* pinned */
    size_t      host_scratch_bytes;
    void       *dev_scratch;
    size_t      dev_scratch_bytes;
    void       *slot[SVT_CTX_SLOTS]; /* grow-only device buffers of the host-pointer convenience entry points */
    size_t      slot_bytes[SVT_CTX_SLOTS];
};

int32_t svt_set_error(int32_t code, const char *msg);
int32_t svt_set_hip_error(hipError_t e, const char *file, int line);
void   *svt_ctx_host_scratch(svt_hip_ctx *ctx, size_t bytes);
void   *svt_ctx_dev_scratch(svt_hip_ctx *ctx, size_t bytes);
void   *svt_ctx_slot(svt_hip_ctx *ctx, int slot, size_t bytes);

#define HIP_TRY(expr)                                                        \
    do {                                                                     \
        hipError_t e_ = (expr);                                              \
        if (e_ != hipSuccess) return svt_set_hip_error(e_, __FILE__, __LINE__); \
    } while (0)
#endif
