/*
 * api.hip -- context management and small host-side helpers of the C ABI (include/svtvp9_hip.h).
 * There is no CPU fallback anywhere in this library: every compute entry point launches HIP kernels
 * and fails with SVT_HIP_ERR_DEVICE when no gfx950 device is usable.
 */
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "svt_ctx.h"

extern "C" void svt_copy_rows_mt(uint8_t *dst, size_t dst_stride, const uint8_t *src, size_t src_stride, size_t width, size_t rows);

static thread_local char g_err[512] = "";

int32_t svt_set_error(int32_t code, const char *msg) {
    snprintf(g_err, sizeof g_err, "%s", msg);
    return code;
}
int32_t svt_set_hip_error(hipError_t e, const char *file, int line) {
    snprintf(g_err, sizeof g_err, "HIP error %d (%s) at %s:%d", (int)e, hipGetErrorString(e), file, line);
    return SVT_HIP_ERR_DEVICE;
}
extern "C" const char *svt_hip_last_error(void) { return g_err; }

extern "C" int32_t svt_hip_sb_count(int32_t w, int32_t h) { return ((w + 63) / 64) * ((h + 63) / 64); }

static int32_t ctx_create(svt_hip_ctx **out, int32_t device, void *stream, int owns, const uint32_t *cu_mask = nullptr, int mask_words = 0) {
    if (!out) return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "ctx: null");
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) return svt_set_error(SVT_HIP_ERR_DEVICE, "ctx: no such HIP device");
    HIP_TRY(hipSetDevice(device));
    svt_hip_ctx *c = (svt_hip_ctx *)calloc(1, sizeof *c);
    if (!c) return svt_set_error(SVT_HIP_ERR_NO_RESOURCES, "ctx: malloc");
    c->device = device;
    {
        hipDeviceProp_t pr;
        c->cu_count = (hipGetDeviceProperties(&pr, device) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256;
    }
    if (owns && cu_mask) {
        if (hipExtStreamCreateWithCUMask(&c->stream, (uint32_t)mask_words, cu_mask) != hipSuccess) { free(c); return svt_set_error(SVT_HIP_ERR_DEVICE, "ctx: CU-masked stream"); }
    } else if (owns) {
        if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { free(c); return svt_set_error(SVT_HIP_ERR_DEVICE, "ctx: stream"); }
    } else {
        c->stream = (hipStream_t)stream;
    }
    c->owns_stream = owns;
    /* on failure release whatever exists: events are created null-checked, destroy tolerates the missing ones */
    bool ok = hipEventCreate(&c->ev_start) == hipSuccess && hipEventCreate(&c->ev_stop) == hipSuccess;
    for (int i = 0; ok && i < SVT_CTX_RING; i++) ok = hipEventCreateWithFlags(&c->ring_ev[i], hipEventDisableTiming) == hipSuccess;
    if (!ok) {
        svt_hip_ctx_destroy(c);
        return svt_set_error(SVT_HIP_ERR_DEVICE, "ctx: events");
    }
    *out = c;
    return SVT_HIP_OK;
}
extern "C" int32_t svt_hip_ctx_create(svt_hip_ctx **ctx, int32_t device) { return ctx_create(ctx, device, nullptr, 1); }
extern "C" int32_t svt_hip_ctx_create_on_stream(svt_hip_ctx **ctx, int32_t device, void *s) { return ctx_create(ctx, device, s, 0); }

extern "C" int32_t svt_hip_ctx_create_cu_mask(svt_hip_ctx **ctx, int32_t device, const uint32_t *cu_mask, int32_t mask_words) {
    if (!cu_mask || mask_words < 1) return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "ctx: empty CU mask");
    return ctx_create(ctx, device, nullptr, 1, cu_mask, mask_words);
}
extern "C" void *svt_hip_ctx_stream(svt_hip_ctx *c) { return c ? (void *)c->stream : nullptr; }

extern "C" void svt_hip_ctx_destroy(svt_hip_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream || !c->owns_stream) (void)hipStreamSynchronize(c->stream);
    for (int i = 0; i < SVT_CTX_SLOTS; i++) if (c->slot[i]) (void)hipFree(c->slot[i]);
    for (int i = 0; i < SVT_CTX_RING; i++) {
        if (c->ring_dev[i]) (void)hipFree(c->ring_dev[i]);
        if (c->ring_host[i]) (void)hipHostFree(c->ring_host[i]);
        if (c->ring_ev[i]) (void)hipEventDestroy(c->ring_ev[i]);
    }
    for (int i = 0; i < SVT_CTX_UPLOAD_RING; i++) {
        if (c->up_host[i]) (void)hipHostFree(c->up_host[i]);
        if (c->up_ev[i]) (void)hipEventDestroy(c->up_ev[i]);
    }
    for (int i = 0; i < SVT_CTX_MARKERS; i++) if (c->mk_ev[i]) (void)hipEventDestroy(c->mk_ev[i]);
    if (c->ho_produced) (void)hipEventDestroy(c->ho_produced);
    if (c->ho_consumed) (void)hipEventDestroy(c->ho_consumed);
    if (c->aux_fork) (void)hipEventDestroy(c->aux_fork);
    for (int i = 0; i < 3; i++) {
        if (c->aux[i]) { (void)hipStreamSynchronize(c->aux[i]); (void)hipStreamDestroy(c->aux[i]); }
        if (c->aux_join[i]) (void)hipEventDestroy(c->aux_join[i]);
    }
    if (c->ev_start) (void)hipEventDestroy(c->ev_start);
    if (c->ev_stop) (void)hipEventDestroy(c->ev_stop);
    if (c->owns_stream && c->stream) (void)hipStreamDestroy(c->stream);
    free(c);
}
int svt_ctx_aux_init(svt_hip_ctx *c) {
    if (c->aux_ready) return 0;
    int prio = 0;
    (void)hipStreamGetPriority(c->stream, &prio);
    /* idempotent per resource: a call that failed half-way leaves its handles in place and the next call creates only what is missing */
    if (!c->aux_fork && hipEventCreateWithFlags(&c->aux_fork, hipEventDisableTiming) != hipSuccess) { c->aux_fork = nullptr; return -1; }
    for (int i = 0; i < 3; i++) {
        if (!c->aux[i] && hipStreamCreateWithPriority(&c->aux[i], hipStreamNonBlocking, prio) != hipSuccess) { c->aux[i] = nullptr; return -1; }
        if (!c->aux_join[i] && hipEventCreateWithFlags(&c->aux_join[i], hipEventDisableTiming) != hipSuccess) { c->aux_join[i] = nullptr; return -1; }
    }
    c->aux_ready = 1;
    return 0;
}

extern "C" int32_t svt_hip_ctx_synchronize(svt_hip_ctx *c) {
    if (!c) return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "ctx: null");
    HIP_TRY(hipStreamSynchronize(c->stream));
    return SVT_HIP_OK;
}
extern "C" float svt_hip_last_kernel_ms(svt_hip_ctx *c) {
    float ms = -1.f;
    if (!c || !c->timed) return ms;
    if (hipEventSynchronize(c->ev_stop) != hipSuccess) return -1.f;
    if (hipEventElapsedTime(&ms, c->ev_start, c->ev_stop) != hipSuccess) return -1.f;
    return ms;
}

int svt_ctx_stage(svt_hip_ctx *c, size_t bytes, void **host, void **dev) {
    const int s = c->ring_pos;
    if (c->ring_used[s]) { if (hipEventSynchronize(c->ring_ev[s]) != hipSuccess) return -1; c->ring_used[s] = 0; }
    if (bytes > c->ring_bytes[s]) {
        if (c->ring_host[s]) (void)hipHostFree(c->ring_host[s]);
        if (c->ring_dev[s]) (void)hipFree(c->ring_dev[s]);
        c->ring_host[s] = nullptr; c->ring_dev[s] = nullptr; c->ring_bytes[s] = 0;
        size_t cap = bytes < 65536 ? 65536 : bytes * 2;
        if (hipHostMalloc(&c->ring_host[s], cap, hipHostMallocDefault) != hipSuccess) return -1;
        if (hipMalloc(&c->ring_dev[s], cap) != hipSuccess) return -1;
        c->ring_bytes[s] = cap;
    }
    *host = c->ring_host[s];
    *dev  = c->ring_dev[s];
    return 0;
}
void svt_ctx_stage_commit(svt_hip_ctx *c) {
    const int s = c->ring_pos;
    (void)hipEventRecord(c->ring_ev[s], c->stream);
    c->ring_used[s] = 1;
    c->ring_pos = (s + 1) % SVT_CTX_RING;
}
void *svt_ctx_slot(svt_hip_ctx *c, int s, size_t bytes) {
    if (s < 0 || s >= SVT_CTX_SLOTS) return nullptr;
    if (bytes > c->slot_bytes[s]) {
        (void)hipStreamSynchronize(c->stream);
        if (c->slot[s]) (void)hipFree(c->slot[s]);
        c->slot[s] = nullptr; c->slot_bytes[s] = 0;
        if (hipMalloc(&c->slot[s], bytes) != hipSuccess) return nullptr;
        c->slot_bytes[s] = bytes;
    }
    return c->slot[s];
}

/* ---- device memory for C hosts ---- */
extern "C" int32_t svt_hip_mem_alloc(svt_hip_ctx *ctx, size_t bytes, void **d_ptr) {
    if (!ctx || !d_ptr || !bytes) return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "mem_alloc: bad argument");
    HIP_TRY(hipSetDevice(ctx->device));
    if (hipMalloc(d_ptr, bytes) != hipSuccess) { *d_ptr = nullptr; return svt_set_error(SVT_HIP_ERR_NO_RESOURCES, "mem_alloc: out of device memory"); }
    return SVT_HIP_OK;
}
extern "C" void svt_hip_mem_free(svt_hip_ctx *ctx, void *d_ptr) {
    if (!ctx || !d_ptr) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_ptr);
}
extern "C" int32_t svt_hip_mem_upload_2d(svt_hip_ctx *ctx, void *d_dst, size_t dst_stride, const void *src, size_t src_stride,
                                         size_t width_bytes, size_t rows) {
    if (!ctx || !d_dst || !src || !width_bytes || !rows || dst_stride < width_bytes || src_stride < width_bytes)
        return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "mem_upload: bad argument");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMemcpy2DAsync(d_dst, dst_stride, src, src_stride, width_bytes, rows, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream)); /* the host rows may be pageable and are the caller's to reuse on return */
    return SVT_HIP_OK;
}
/* The rows are copied into a pinned staging buffer of the context before the call returns (the copy the reference makes of an
 * input picture inside eb_vp9_svt_enc_send_picture, Codec/EbEncHandle.c:2743-2796): the caller may reuse them at once, the
 * host-to-device copy runs asynchronously in stream order. */
extern "C" int32_t svt_hip_mem_upload_2d_async(svt_hip_ctx *ctx, void *d_dst, size_t dst_stride, const void *src, size_t src_stride,
                                               size_t width_bytes, size_t rows) {
    if (!ctx || !d_dst || !src || !width_bytes || !rows || dst_stride < width_bytes || src_stride < width_bytes)
        return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "mem_upload_async: bad argument");
    HIP_TRY(hipSetDevice(ctx->device));
    const int    k = ctx->up_pos;
    const size_t bytes = width_bytes * rows;
    if (ctx->up_used[k]) { HIP_TRY(hipEventSynchronize(ctx->up_ev[k])); ctx->up_used[k] = 0; }
    if (!ctx->up_ev[k]) HIP_TRY(hipEventCreateWithFlags(&ctx->up_ev[k], hipEventDisableTiming));
    if (bytes > ctx->up_bytes[k]) {
        if (ctx->up_host[k]) (void)hipHostFree(ctx->up_host[k]);
        ctx->up_host[k] = nullptr; ctx->up_bytes[k] = 0;
        if (hipHostMalloc(&ctx->up_host[k], bytes, hipHostMallocDefault) != hipSuccess) return svt_set_error(SVT_HIP_ERR_NO_RESOURCES, "mem_upload_async: pinned staging buffer");
        ctx->up_bytes[k] = bytes;
    }
    uint8_t *st = (uint8_t *)ctx->up_host[k];
    svt_copy_rows_mt(st, width_bytes, (const uint8_t *)src, src_stride, width_bytes, rows); /* host/copy_pool.c: a few threads */
    /* the staging buffer is tight: a contiguous destination takes one linear copy (the DMA engines' fast path) */
    if (dst_stride == width_bytes) HIP_TRY(hipMemcpyAsync(d_dst, st, bytes, hipMemcpyHostToDevice, ctx->stream));
    else HIP_TRY(hipMemcpy2DAsync(d_dst, dst_stride, st, width_bytes, width_bytes, rows, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipEventRecord(ctx->up_ev[k], ctx->stream));
    ctx->up_used[k] = 1;
    ctx->up_pos = (k + 1) % SVT_CTX_UPLOAD_RING;
    return SVT_HIP_OK;
}

/* ---- completion markers: "everything enqueued on the context's stream so far" as a value a host can poll or wait for ---- */
extern "C" int32_t svt_hip_ctx_marker_record(svt_hip_ctx *ctx, uint64_t *marker) {
    if (!ctx || !marker) return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "marker: null");
    HIP_TRY(hipSetDevice(ctx->device));
    const uint64_t m = ctx->mk_next;
    hipEvent_t    &e = ctx->mk_ev[m % SVT_CTX_MARKERS];
    if (!e) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    else HIP_TRY(hipEventSynchronize(e)); /* its previous use (marker m - SVT_CTX_MARKERS) is complete from here on */
    HIP_TRY(hipEventRecord(e, ctx->stream));
    ctx->mk_next = m + 1;
    *marker = m;
    return SVT_HIP_OK;
}
extern "C" int32_t svt_hip_ctx_marker_query(svt_hip_ctx *ctx, uint64_t marker) {
    if (!ctx || marker >= ctx->mk_next) return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "marker: unknown");
    if (ctx->mk_next - marker > SVT_CTX_MARKERS) return 1; /* its event has been waited for and recorded again since */
    const hipError_t e = hipEventQuery(ctx->mk_ev[marker % SVT_CTX_MARKERS]);
    if (e == hipSuccess) return 1;
    if (e == hipErrorNotReady) { (void)hipGetLastError(); return 0; }
    return svt_set_hip_error(e, __FILE__, __LINE__);
}
extern "C" int32_t svt_hip_ctx_marker_wait(svt_hip_ctx *ctx, uint64_t marker) {
    if (!ctx || marker >= ctx->mk_next) return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "marker: unknown");
    if (ctx->mk_next - marker > SVT_CTX_MARKERS) return SVT_HIP_OK;
    HIP_TRY(hipEventSynchronize(ctx->mk_ev[marker % SVT_CTX_MARKERS]));
    return SVT_HIP_OK;
}

extern "C" int32_t svt_hip_mem_download(svt_hip_ctx *ctx, void *dst, const void *d_src, size_t bytes) {
    if (!ctx || !dst || !d_src || !bytes) return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "mem_download: bad argument");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return SVT_HIP_OK;
}
extern "C" int32_t svt_hip_mem_set(svt_hip_ctx *ctx, void *d_dst, int32_t value, size_t bytes) {
    if (!ctx || !d_dst || !bytes) return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "mem_set: bad argument");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMemsetAsync(d_dst, value, bytes, ctx->stream));
    return SVT_HIP_OK;
}

/* ---- stream-to-stream ordering between contexts, pinned host memory, asynchronous downloads ---- */
extern "C" int32_t svt_hip_ctx_wait_marker(svt_hip_ctx *ctx, svt_hip_ctx *other, uint64_t marker) {
    if (!ctx || !other || marker >= other->mk_next) return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "wait_marker: unknown");
    if (other->mk_next - marker > SVT_CTX_MARKERS) return SVT_HIP_OK; /* long complete */
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamWaitEvent(ctx->stream, other->mk_ev[marker % SVT_CTX_MARKERS], 0));
    return SVT_HIP_OK;
}
extern "C" int32_t svt_hip_host_alloc(svt_hip_ctx *ctx, size_t bytes, void **ptr) {
    if (!ctx || !ptr || !bytes) return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "host_alloc: bad argument");
    HIP_TRY(hipSetDevice(ctx->device));
    if (hipHostMalloc(ptr, bytes, hipHostMallocDefault) != hipSuccess) { *ptr = nullptr; return svt_set_error(SVT_HIP_ERR_NO_RESOURCES, "host_alloc: pinned memory"); }
    return SVT_HIP_OK;
}
extern "C" void svt_hip_host_free(svt_hip_ctx *ctx, void *ptr) {
    if (!ctx || !ptr) return;
    (void)hipSetDevice(ctx->device);
    (void)hipHostFree(ptr);
}
extern "C" int32_t svt_hip_mem_download_2d_async(svt_hip_ctx *ctx, void *dst, size_t dst_stride, const void *d_src, size_t src_stride, size_t width_bytes,
                                                 size_t rows) {
    if (!ctx || !dst || !d_src || !width_bytes || !rows || dst_stride < width_bytes || src_stride < width_bytes)
        return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "mem_download_2d: bad argument");
    HIP_TRY(hipSetDevice(ctx->device));
    if (dst_stride == width_bytes && src_stride == width_bytes) HIP_TRY(hipMemcpyAsync(dst, d_src, width_bytes * rows, hipMemcpyDeviceToHost, ctx->stream));
    else HIP_TRY(hipMemcpy2DAsync(dst, dst_stride, d_src, src_stride, width_bytes, rows, hipMemcpyDeviceToHost, ctx->stream));
    return SVT_HIP_OK;
}
extern "C" int32_t svt_hip_mem_copy_2d_device(svt_hip_ctx *ctx, void *d_dst, size_t dst_stride, const void *d_src, size_t src_stride, size_t width_bytes,
                                              size_t rows) {
    if (!ctx || !d_dst || !d_src || !width_bytes || !rows || dst_stride < width_bytes || src_stride < width_bytes)
        return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "mem_copy_2d: bad argument");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMemcpy2DAsync(d_dst, dst_stride, d_src, src_stride, width_bytes, rows, hipMemcpyDeviceToDevice, ctx->stream));
    return SVT_HIP_OK;
}
