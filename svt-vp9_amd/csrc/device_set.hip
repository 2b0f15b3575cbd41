/*
 * device_set.hip -- one context per GPU of a node and the device-to-device hand-off of a reference picture (scope row e).
 * A process that drives several GPUs itself (the reference is one multi-threaded C process) uses these; a job that runs one
 * process per GPU (bench.py under torch.distributed) hands the same buffer over with an RCCL send / recv pair instead.
 */
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "svt_ctx.h"

struct svt_hip_device_set {
    int          n;
    svt_hip_ctx **ctx;
};

extern "C" int32_t svt_hip_device_set_create(svt_hip_device_set **out, const int32_t *ordinals, int32_t n) {
    if (!out || !ordinals || n < 1) return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "device_set: bad argument");
    *out = nullptr;
    svt_hip_device_set *s = (svt_hip_device_set *)calloc(1, sizeof *s);
    if (!s) return svt_set_error(SVT_HIP_ERR_NO_RESOURCES, "device_set: malloc");
    s->ctx = (svt_hip_ctx **)calloc((size_t)n, sizeof *s->ctx);
    if (!s->ctx) { free(s); return svt_set_error(SVT_HIP_ERR_NO_RESOURCES, "device_set: malloc"); }
    for (int i = 0; i < n; i++) {
        const int32_t rc = svt_hip_ctx_create(&s->ctx[i], ordinals[i]);
        if (rc) { s->n = i; svt_hip_device_set_destroy(s); return rc; }
    }
    s->n = n;
    *out = s;
    return SVT_HIP_OK;
}
extern "C" int32_t svt_hip_device_set_size(const svt_hip_device_set *s) { return s ? s->n : 0; }
extern "C" svt_hip_ctx *svt_hip_device_set_ctx(svt_hip_device_set *s, int32_t i) { return s && i >= 0 && i < s->n ? s->ctx[i] : nullptr; }
extern "C" void svt_hip_device_set_destroy(svt_hip_device_set *s) {
    if (!s) return;
    for (int i = 0; i < s->n; i++) svt_hip_ctx_destroy(s->ctx[i]);
    free(s->ctx);
    free(s);
}

extern "C" int32_t svt_hip_ref_handoff_device(svt_hip_ctx *src, const void *d_src, svt_hip_ctx *dst, void *d_dst, size_t bytes) {
    if (!src || !dst || !d_src || !d_dst || !bytes) return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "handoff: bad argument");
    if (src->device != dst->device) { /* peer access: once per direction; "already enabled" is not an error */
        int can = 0;
        HIP_TRY(hipDeviceCanAccessPeer(&can, dst->device, src->device));
        if (can) {
            HIP_TRY(hipSetDevice(dst->device));
            const hipError_t e = hipDeviceEnablePeerAccess(src->device, 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) return svt_set_hip_error(e, __FILE__, __LINE__);
            (void)hipGetLastError();
        }
    }
    /* producer's stream -> event -> consumer's stream: the copy runs on the consumer's stream after the producer's work.  The two
     * events are the contexts' own hand-off events (not shared with any other entry point).  The call touches BOTH contexts: the
     * caller serialises it against other calls on either of them (one thread per device: do the hand-off while the consumer's
     * thread is not enqueuing, e.g. at the mini-GOP boundary it synchronises on anyway). */
    HIP_TRY(hipSetDevice(src->device));
    if (!src->ho_produced) HIP_TRY(hipEventCreateWithFlags(&src->ho_produced, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(src->ho_produced, src->stream));
    HIP_TRY(hipSetDevice(dst->device));
    if (!dst->ho_consumed) HIP_TRY(hipEventCreateWithFlags(&dst->ho_consumed, hipEventDisableTiming));
    HIP_TRY(hipStreamWaitEvent(dst->stream, src->ho_produced, 0));
    if (src->device == dst->device) HIP_TRY(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, dst->stream));
    else HIP_TRY(hipMemcpyPeerAsync(d_dst, dst->device, d_src, src->device, bytes, dst->stream));
    /* the producer may reuse its buffer only once the copy has read it */
    HIP_TRY(hipEventRecord(dst->ho_consumed, dst->stream));
    HIP_TRY(hipSetDevice(src->device));
    HIP_TRY(hipStreamWaitEvent(src->stream, dst->ho_consumed, 0));
    return SVT_HIP_OK;
}
