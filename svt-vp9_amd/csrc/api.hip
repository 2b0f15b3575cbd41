/*
 * api.hip -- context management and small host-side helpers of the C ABI (include/svtvp9_hip.h).
 * There is no CPU fallback anywhere in this library: every compute entry point launches HIP kernels
 * and fails with SVT_HIP_ERR_DEVICE when no gfx950 device is usable.
 */
#include <hip/hip_runtime.h>
#include <time.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "svt_ctx.h"

extern "C" void svt_copy_rows_mt(uint8_t *dst, size_t dst_stride, const uint8_t *src, size_t src_stride, size_t width, size_t rows);
extern "C" void svt_copy_planes_mt(int n_planes, uint8_t *const *dst, const size_t *dst_stride, const uint8_t *const *src, const size_t *src_stride, const size_t *width,
                                   const size_t *rows);

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
    if (ok) { /* the staging ring's buffers in two slabs (svt_ctx_stage grows an entry that needs more than its share) */
        const size_t unit = 65536;
        if (hipHostMalloc(&c->ring_slab_host, unit * SVT_CTX_RING, hipHostMallocDefault) == hipSuccess && hipMalloc(&c->ring_slab_dev, unit * SVT_CTX_RING) == hipSuccess)
            for (int i = 0; i < SVT_CTX_RING; i++) {
                c->ring_host[i] = (uint8_t *)c->ring_slab_host + unit * i; c->ring_dev[i] = (uint8_t *)c->ring_slab_dev + unit * i; c->ring_bytes[i] = unit;
            }
        else { /* (not fatal: the entries allocate on demand) */
            if (c->ring_slab_host) (void)hipHostFree(c->ring_slab_host);
            c->ring_slab_host = nullptr; c->ring_slab_dev = nullptr;
        }
    }
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
/* Make the context's stream real before a clock starts: the runtime creates a stream's hardware queue the first time work is submitted
 * to it, and sizes the queue's scratch memory the first time a kernel with a private segment runs on it -- for the intra pass's kernel
 * (752 bytes per lane: ~400 MB for a device's worth of waves) that was 4 ms on the thread that sent the first key frame.  scratch_bytes: the
 * largest private segment (per lane) of the kernels the stream will run; 0: none. */
template <int WORDS> __global__ void svt_ctx_warm_kernel(uint32_t *p, int n) {
    if constexpr (WORDS > 0) {
        volatile uint32_t a[WORDS];
        uint32_t          acc = 0;
        for (int i = 0; i < n && i < WORDS; i++) a[i] = (uint32_t)(i * n);
        for (int i = 0; i < n; i++) acc += a[(i * 7) % WORDS];
        if (p && n > 0) *p = acc;
    } else if (p && n > 0) *p = 0;
}
extern "C" int32_t svt_hip_ctx_warm_scratch(svt_hip_ctx *c, int32_t scratch_bytes) {
    if (!c || scratch_bytes < 0) return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "ctx_warm: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    uint32_t *p = (uint32_t *)c->ring_slab_dev;
    /* (a grid that fills the device: the runtime sizes a queue's scratch by the dispatch that asks for it) */
    const dim3 grid((unsigned)(c->cu_count > 0 ? c->cu_count * 8 : 2048));
    if (scratch_bytes == 0) hipLaunchKernelGGL(svt_ctx_warm_kernel<0>, dim3(1), dim3(64), 0, c->stream, p, 0);
    else if (scratch_bytes <= 256) hipLaunchKernelGGL(svt_ctx_warm_kernel<64>, grid, dim3(64), 0, c->stream, p, 0);
    else hipLaunchKernelGGL(svt_ctx_warm_kernel<256>, grid, dim3(64), 0, c->stream, p, 0);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    return SVT_HIP_OK;
}
extern "C" int32_t svt_hip_ctx_warm(svt_hip_ctx *c) { return svt_hip_ctx_warm_scratch(c, 0); }
extern "C" int32_t svt_hip_ctx_set_intra_workgroups(svt_hip_ctx *c, int32_t n) {
    if (!c || n < 0) return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "ctx: intra workgroups");
    c->intra_wgs = n;
    return SVT_HIP_OK;
}

extern "C" void svt_hip_ctx_destroy(svt_hip_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->up_prof_n)
        fprintf(stderr, "svt_hip upload profile (us per picture): staging-slot wait %.1f  row copies %.1f  enqueue %.1f  (%ld uploads)\n", 1e6 * c->up_prof[0] / (double)c->up_prof_n,
                1e6 * c->up_prof[1] / (double)c->up_prof_n, 1e6 * c->up_prof[2] / (double)c->up_prof_n, c->up_prof_n);
    if (c->stream || !c->owns_stream) (void)hipStreamSynchronize(c->stream);
    for (int i = 0; i < SVT_CTX_SLOTS; i++) if (c->slot[i]) (void)hipFree(c->slot[i]);
    for (int i = 0; i < SVT_CTX_RING; i++) {
        if (c->ring_own[i]) {
            if (c->ring_dev[i]) (void)hipFree(c->ring_dev[i]);
            if (c->ring_host[i]) (void)hipHostFree(c->ring_host[i]);
        }
        if (c->ring_ev[i]) (void)hipEventDestroy(c->ring_ev[i]);
    }
    if (c->ring_slab_dev) (void)hipFree(c->ring_slab_dev);
    if (c->ring_slab_host) (void)hipHostFree(c->ring_slab_host);
    for (int i = 0; i < SVT_CTX_UPLOAD_RING; i++) {
        if (c->up_host[i]) (void)hipHostFree(c->up_host[i]);
        if (c->up_ev[i]) (void)hipEventDestroy(c->up_ev[i]);
        if (i == 0 && c->direct_ev) (void)hipEventDestroy(c->direct_ev);
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
        if (c->ring_own[s]) { /* (an entry inside the slabs owns nothing) */
            if (c->ring_host[s]) (void)hipHostFree(c->ring_host[s]);
            if (c->ring_dev[s]) (void)hipFree(c->ring_dev[s]);
        }
        c->ring_host[s] = nullptr; c->ring_dev[s] = nullptr; c->ring_bytes[s] = 0; c->ring_own[s] = 1;
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
/* flags of the pinned staging buffers (experiment aid: SVT_HIP_STAGING_FLAGS = a hipHostMalloc flag word) */
static unsigned svt_staging_flags() {
    static const unsigned f = [] { const char *e = getenv("SVT_HIP_STAGING_FLAGS"); return e ? (unsigned)strtoul(e, nullptr, 0) : (unsigned)hipHostMallocDefault; }();
    return f;
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
        if (hipHostMalloc(&ctx->up_host[k], bytes, svt_staging_flags()) != hipSuccess) return svt_set_error(SVT_HIP_ERR_NO_RESOURCES, "mem_upload_async: pinned staging buffer");
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

/* The planes of one picture in one go: staged back to back in ONE slot of the ring; destinations that lie back to back on the device,
 * tight (a picture's Y | Cb | Cr inside one buffer), take ONE host-to-device copy, and there is one event for the slot -- a third of the
 * stream operations of three svt_hip_mem_upload_2d_async calls (the public API's picture rate is bound by the number of operations its
 * thread enqueues). */
extern "C" int32_t svt_hip_mem_upload_planes_async(svt_hip_ctx *ctx, int32_t n_planes, void *const *d_dst, const size_t *dst_stride, const void *const *src,
                                                   const size_t *src_stride, const size_t *width_bytes, const size_t *rows) {
    if (!ctx || n_planes < 1 || n_planes > 4 || !d_dst || !dst_stride || !src || !src_stride || !width_bytes || !rows)
        return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "mem_upload_planes: bad argument");
    size_t bytes = 0;
    bool   one = true;
    for (int i = 0; i < n_planes; i++) {
        if (!d_dst[i] || !src[i] || !width_bytes[i] || !rows[i] || dst_stride[i] < width_bytes[i] || src_stride[i] < width_bytes[i])
            return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "mem_upload_planes: bad plane");
        if (dst_stride[i] != width_bytes[i] || (i && (uint8_t *)d_dst[i] != (uint8_t *)d_dst[i - 1] + width_bytes[i - 1] * rows[i - 1])) one = false;
        bytes += width_bytes[i] * rows[i];
    }
    HIP_TRY(hipSetDevice(ctx->device));
    static const bool prof = getenv("SVT_HIP_SHIM_PROFILE") != nullptr && atoi(getenv("SVT_HIP_SHIM_PROFILE")) != 0;
    auto now = [] { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; };
    double t0 = prof ? now() : 0.0;
    const int k = ctx->up_pos;
    if (ctx->up_used[k]) { HIP_TRY(hipEventSynchronize(ctx->up_ev[k])); ctx->up_used[k] = 0; }
    if (!ctx->up_ev[k]) HIP_TRY(hipEventCreateWithFlags(&ctx->up_ev[k], hipEventDisableTiming));
    if (bytes > ctx->up_bytes[k]) {
        if (ctx->up_host[k]) (void)hipHostFree(ctx->up_host[k]);
        ctx->up_host[k] = nullptr; ctx->up_bytes[k] = 0;
        if (hipHostMalloc(&ctx->up_host[k], bytes, svt_staging_flags()) != hipSuccess) return svt_set_error(SVT_HIP_ERR_NO_RESOURCES, "mem_upload_planes: pinned staging buffer");
        ctx->up_bytes[k] = bytes;
    }
    if (prof) { const double t1 = now(); ctx->up_prof[0] += t1 - t0; t0 = t1; }
    uint8_t *st = (uint8_t *)ctx->up_host[k];
    size_t   off = 0;
    {   /* host/copy_pool.c: a few threads, one hand-over for the whole picture */
        uint8_t *sd[4];
        for (int i = 0; i < n_planes; i++) { sd[i] = st + off; off += width_bytes[i] * rows[i]; }
        svt_copy_planes_mt(n_planes, sd, width_bytes, (const uint8_t *const *)src, src_stride, width_bytes, rows);
    }
    if (prof) { const double t1 = now(); ctx->up_prof[1] += t1 - t0; t0 = t1; }
    if (one) HIP_TRY(hipMemcpyAsync(d_dst[0], st, bytes, hipMemcpyHostToDevice, ctx->stream));
    else {
        off = 0;
        for (int i = 0; i < n_planes; i++) {
            if (dst_stride[i] == width_bytes[i]) HIP_TRY(hipMemcpyAsync(d_dst[i], st + off, width_bytes[i] * rows[i], hipMemcpyHostToDevice, ctx->stream));
            else HIP_TRY(hipMemcpy2DAsync(d_dst[i], dst_stride[i], st + off, width_bytes[i], width_bytes[i], rows[i], hipMemcpyHostToDevice, ctx->stream));
            off += width_bytes[i] * rows[i];
        }
    }
    HIP_TRY(hipEventRecord(ctx->up_ev[k], ctx->stream));
    ctx->up_used[k] = 1;
    ctx->up_pos = (k + 1) % SVT_CTX_UPLOAD_RING;
    if (prof) { ctx->up_prof[2] += now() - t0; ctx->up_prof_n++; }
    return SVT_HIP_OK;
}

/* ---- uploads straight from the caller's memory (opt-in) ----
 * A host that sends its pictures from a fixed pool of buffers which stay allocated for the encoder's lifetime (the reference's
 * application does: allocate_input_buffers, App/EbAppContext.c) does not need the staging copy: the first time a range of host
 * memory is seen it is page-locked (hipHostRegister, portable across devices), from then on the DMA engines read it directly at
 * the link's rate -- measured on the MI355X box: 57 GB/s from pinned memory against the ~27 GB/s a staging memcpy by a few threads
 * sustains under the box's CPU quota.  The registry keeps disjoint page-aligned intervals (a picture's planes usually share pages
 * with their neighbours: only the uncovered part of a new range is registered) and a copy is cut where two intervals meet (the
 * runtime resolves a host pointer to ONE registration and rejects a copy that runs past its end).  Registration stops after
 * SVT_REG_MAX intervals; a range that cannot be registered is served by the staging path (svt_hip_mem_upload_2d_async), so the
 * result is the same either way.  Page-locking memory the library does not own is the CALLER's promise that it stays allocated
 * (and is not handed, in part, to other copies of this process while locked): the encoder library only takes this path when
 * SVT_HIP_REGISTER_INPUT=1.  The copy is asynchronous; svt_hip_mem_upload_wait() returns when every direct upload of the context
 * has left the caller's memory -- the point where eb_vp9_svt_enc_send_picture may return. */
#include <mutex>
#include <unistd.h>
namespace {
#define SVT_REG_MAX 1024
struct reg_iv { uintptr_t lo, hi; bool ok; };
std::mutex g_reg_mutex;
reg_iv     g_reg[SVT_REG_MAX];
int        g_reg_n = 0;
bool       g_reg_off = false;
int        g_reg_users = 0; /* encoders (or other hosts) that hold the registry: svt_hip_host_registry_retain / _release */

/* end of the registered interval that holds address a (a is covered: reg_cover succeeded for it) */
uintptr_t reg_end_of(uintptr_t a) {
    std::lock_guard<std::mutex> lock(g_reg_mutex);
    for (int i = 0; i < g_reg_n; i++) if (g_reg[i].lo <= a && a < g_reg[i].hi) return g_reg[i].hi;
    return a;
}
/* true when [a, b) is page-locked after the call */
bool reg_cover(uintptr_t a, uintptr_t b) {
    static const uintptr_t page = (uintptr_t)sysconf(_SC_PAGESIZE);
    const uintptr_t A = a & ~(page - 1), B = (b + page - 1) & ~(page - 1);
    std::lock_guard<std::mutex> lock(g_reg_mutex);
    bool      all_ok = true;
    uintptr_t cur = A;
    while (cur < B) {
        const reg_iv *in = nullptr; /* the interval that covers `cur`, or the start of the next one above it */
        uintptr_t     next = B;
        for (int i = 0; i < g_reg_n; i++) {
            if (g_reg[i].lo <= cur && cur < g_reg[i].hi) { in = &g_reg[i]; break; }
            if (g_reg[i].lo > cur && g_reg[i].lo < next) next = g_reg[i].lo;
        }
        if (in) { all_ok = all_ok && in->ok; cur = in->hi; continue; }
        if (g_reg_off || g_reg_n == SVT_REG_MAX) { g_reg_off = true; return false; }
        const hipError_t e = hipHostRegister((void *)cur, next - cur, hipHostRegisterPortable);
        (void)hipGetLastError();
        /* a range that could not be locked is NOT remembered: this upload goes through the staging path and a later one tries again (the
           refusal may have been transient -- locked-memory limit, a mapping that has changed since) */
        if (e == hipSuccess) g_reg[g_reg_n++] = reg_iv{cur, next, true};
        else all_ok = false;
        cur = next;
    }
    return all_ok;
}
} // namespace

extern "C" int32_t svt_hip_mem_upload_2d_direct(svt_hip_ctx *ctx, void *d_dst, size_t dst_stride, const void *src, size_t src_stride,
                                                size_t width_bytes, size_t rows) {
    if (!ctx || !d_dst || !src || !width_bytes || !rows || dst_stride < width_bytes || src_stride < width_bytes)
        return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "mem_upload_direct: bad argument");
    HIP_TRY(hipSetDevice(ctx->device));
    if (!reg_cover((uintptr_t)src, (uintptr_t)src + src_stride * (rows - 1) + width_bytes))
        return svt_hip_mem_upload_2d_async(ctx, d_dst, dst_stride, src, src_stride, width_bytes, rows);
    const uint8_t *sp = (const uint8_t *)src;
    uint8_t       *dp = (uint8_t *)d_dst;
    if (dst_stride == width_bytes && src_stride == width_bytes) { /* contiguous: one run of bytes, cut at the registry's interval ends */
        size_t left = width_bytes * rows;
        while (left) {
            const uintptr_t end = reg_end_of((uintptr_t)sp);
            const size_t    n = (size_t)(end - (uintptr_t)sp) < left ? (size_t)(end - (uintptr_t)sp) : left;
            if (!n) return svt_set_error(SVT_HIP_ERR_DEVICE, "mem_upload_direct: registry");
            HIP_TRY(hipMemcpyAsync(dp, sp, n, hipMemcpyHostToDevice, ctx->stream));
            sp += n; dp += n; left -= n;
        }
    } else {
        size_t r = 0;
        while (r < rows) {
            const uint8_t  *row = sp + r * src_stride;
            const uintptr_t end = reg_end_of((uintptr_t)row);
            if ((uintptr_t)row + width_bytes <= end) { /* this row and maybe more lie inside the interval */
                size_t nfull = (size_t)(end - (uintptr_t)row - width_bytes) / src_stride + 1;
                if (nfull > rows - r) nfull = rows - r;
                HIP_TRY(hipMemcpy2DAsync(dp + r * dst_stride, dst_stride, row, src_stride, width_bytes, nfull, hipMemcpyHostToDevice, ctx->stream));
                r += nfull;
            } else { /* the row straddles a boundary: in pieces */
                size_t off = 0;
                while (off < width_bytes) {
                    const uintptr_t e2 = reg_end_of((uintptr_t)row + off);
                    const size_t    n = (size_t)(e2 - ((uintptr_t)row + off)) < width_bytes - off ? (size_t)(e2 - ((uintptr_t)row + off)) : width_bytes - off;
                    if (!n) return svt_set_error(SVT_HIP_ERR_DEVICE, "mem_upload_direct: registry");
                    HIP_TRY(hipMemcpyAsync(dp + r * dst_stride + off, row + off, n, hipMemcpyHostToDevice, ctx->stream));
                    off += n;
                }
                r++;
            }
        }
    }
    if (!ctx->direct_ev) HIP_TRY(hipEventCreateWithFlags(&ctx->direct_ev, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(ctx->direct_ev, ctx->stream));
    ctx->direct_pending = 1;
    return SVT_HIP_OK;
}
extern "C" int32_t svt_hip_mem_upload_wait(svt_hip_ctx *ctx) {
    if (!ctx) return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "mem_upload_wait: null");
    if (ctx->direct_pending) { HIP_TRY(hipEventSynchronize(ctx->direct_ev)); ctx->direct_pending = 0; }
    return SVT_HIP_OK;
}
extern "C" void svt_hip_host_unregister_all(void) {
    std::lock_guard<std::mutex> lock(g_reg_mutex);
    for (int i = 0; i < g_reg_n; i++) if (g_reg[i].ok) (void)hipHostUnregister((void *)g_reg[i].lo);
    g_reg_n = 0; g_reg_off = false;
}
/* The registry is process-wide (a host range is locked once, whichever context uploads from it); its life is tied to its users: a host
 * that uploads with svt_hip_mem_upload_2d_direct retains it while it is live and releases it when every upload of its own has completed.
 * The last release unlocks every range, so that memory the application frees afterwards -- and a later allocation at the same address --
 * is never read through a stale registration. */
extern "C" void svt_hip_host_registry_retain(void) {
    std::lock_guard<std::mutex> lock(g_reg_mutex);
    g_reg_users++;
}
extern "C" void svt_hip_host_registry_release(void) {
    std::lock_guard<std::mutex> lock(g_reg_mutex);
    if (g_reg_users > 0 && --g_reg_users == 0) {
        for (int i = 0; i < g_reg_n; i++) if (g_reg[i].ok) (void)hipHostUnregister((void *)g_reg[i].lo);
        g_reg_n = 0; g_reg_off = false;
    }
}

/* ---- completion markers: "everything enqueued on the context's stream so far" as a value a host can poll or wait for ---- */
extern "C" int32_t svt_hip_ctx_marker_record(svt_hip_ctx *ctx, uint64_t *marker) {
    if (!ctx || !marker) return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "marker: null");
    HIP_TRY(hipSetDevice(ctx->device));
    const uint64_t m = ctx->mk_next; /* (only the context's enqueuing thread records; other threads may query / wait: atomic counter) */
    hipEvent_t    &e = ctx->mk_ev[m % SVT_CTX_MARKERS];
    if (!e) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    else HIP_TRY(hipEventSynchronize(e)); /* its previous use (marker m - SVT_CTX_MARKERS) is complete from here on */
    HIP_TRY(hipEventRecord(e, ctx->stream));
    __atomic_store_n(&ctx->mk_next, m + 1, __ATOMIC_RELEASE);
    *marker = m;
    return SVT_HIP_OK;
}
extern "C" int32_t svt_hip_ctx_marker_query(svt_hip_ctx *ctx, uint64_t marker) {
    const uint64_t next = ctx ? __atomic_load_n(&ctx->mk_next, __ATOMIC_ACQUIRE) : 0;
    if (!ctx || marker >= next) return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "marker: unknown");
    if (next - marker > SVT_CTX_MARKERS) return 1; /* its event has been waited for and recorded again since */
    const hipError_t e = hipEventQuery(ctx->mk_ev[marker % SVT_CTX_MARKERS]);
    if (e == hipSuccess) return 1;
    if (e == hipErrorNotReady) { (void)hipGetLastError(); return 0; }
    return svt_set_hip_error(e, __FILE__, __LINE__);
}
extern "C" int32_t svt_hip_ctx_marker_wait(svt_hip_ctx *ctx, uint64_t marker) {
    const uint64_t next = ctx ? __atomic_load_n(&ctx->mk_next, __ATOMIC_ACQUIRE) : 0;
    if (!ctx || marker >= next) return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "marker: unknown");
    if (next - marker > SVT_CTX_MARKERS) return SVT_HIP_OK;
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
    const uint64_t next = other ? __atomic_load_n(&other->mk_next, __ATOMIC_ACQUIRE) : 0;
    if (!ctx || !other || marker >= next) return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "wait_marker: unknown");
    if (next - marker > SVT_CTX_MARKERS) return SVT_HIP_OK; /* long complete */
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
