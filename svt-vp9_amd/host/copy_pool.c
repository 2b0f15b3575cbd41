/*
 * copy_pool.c -- the host copy of an input picture into pinned staging (svt_hip_mem_upload_2d_async = the copy
 * eb_vp9_svt_enc_send_picture makes of the caller's picture, Source/Lib/Codec/EbEncHandle.c:2743-2796), spread over a few
 * threads: one thread moves ~10-25 GB/s, a 4K luma plane is 8.3 MB, and that copy -- not the PCIe transfer behind it, not the GPU --
 * is what bounds the public API's picture rate.  The reference's own input path is multi-threaded as well (resource coordination
 * and picture analysis run in their own threads).  A process-wide fork-join pool, created on first use; SVT_HIP_COPY_THREADS
 * (default 4, 1 = copy in the calling thread).  Plain C, pthreads.
 * Lifetime: the workers are joinable and are stopped and joined when the library is unloaded (destructor below), so a dlclose
 * leaves no thread running on unmapped code; a fork() child starts without workers (they do not exist there) and copies in the
 * calling thread.  One fork-join at a time: callers of different contexts take turns (a copy saturates the memory controllers
 * of its socket anyway).
 */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define POOL_MAX 16
#define POOL_PLANES 4

static struct {
    pthread_mutex_t lock;      /* protects the job fields and the counters */
    pthread_cond_t  go, done;
    pthread_mutex_t busy;      /* one fork-join at a time (callers of different contexts may arrive together) */
    pthread_t       th[POOL_MAX];
    int             n;         /* workers besides the caller */
    unsigned long   gen;       /* job generation */
    int             stop;      /* set by the destructor: workers leave */
    int             pending;
    int             n_planes;  /* the job: up to POOL_PLANES planes, every thread takes its share of the rows of each */
    uint8_t        *dst[POOL_PLANES];
    const uint8_t  *src[POOL_PLANES];
    size_t          dst_stride[POOL_PLANES], src_stride[POOL_PLANES], width[POOL_PLANES], rows[POOL_PLANES];
} g_pool = {.lock = PTHREAD_MUTEX_INITIALIZER, .go = PTHREAD_COND_INITIALIZER, .done = PTHREAD_COND_INITIALIZER, .busy = PTHREAD_MUTEX_INITIALIZER};
static pthread_once_t g_once = PTHREAD_ONCE_INIT;

static void copy_rows(uint8_t *dst, size_t dst_stride, const uint8_t *src, size_t src_stride, size_t width, size_t r0, size_t r1) {
    if (src_stride == width && dst_stride == width) memcpy(dst + r0 * width, src + r0 * width, (r1 - r0) * width);
    else for (size_t r = r0; r < r1; r++) memcpy(dst + r * dst_stride, src + r * src_stride, width);
}
static void copy_slice(int part, int parts) {
    for (int p = 0; p < g_pool.n_planes; p++) {
        const size_t r0 = g_pool.rows[p] * (size_t)part / (size_t)parts, r1 = g_pool.rows[p] * (size_t)(part + 1) / (size_t)parts;
        copy_rows(g_pool.dst[p], g_pool.dst_stride[p], g_pool.src[p], g_pool.src_stride[p], g_pool.width[p], r0, r1);
    }
}

static void *worker(void *arg) {
    const int     id = (int)(intptr_t)arg;
    unsigned long seen = 0;
    for (;;) {
        pthread_mutex_lock(&g_pool.lock);
        while (g_pool.gen == seen && !g_pool.stop) pthread_cond_wait(&g_pool.go, &g_pool.lock);
        if (g_pool.stop) { pthread_mutex_unlock(&g_pool.lock); break; }
        seen = g_pool.gen;
        pthread_mutex_unlock(&g_pool.lock);
        copy_slice(id + 1, g_pool.n + 1);
        pthread_mutex_lock(&g_pool.lock);
        if (--g_pool.pending == 0) pthread_cond_signal(&g_pool.done);
        pthread_mutex_unlock(&g_pool.lock);
    }
    return NULL;
}

/* in a fork() child the workers do not exist: forget them (the locks are re-initialised: no other thread survived the fork) */
static void pool_atfork_child(void) {
    g_pool.n = 0;
    g_pool.pending = 0;
    pthread_mutex_init(&g_pool.lock, NULL);
    pthread_mutex_init(&g_pool.busy, NULL);
    pthread_cond_init(&g_pool.go, NULL);
    pthread_cond_init(&g_pool.done, NULL);
}

/* library unload / process exit: stop and join the workers */
__attribute__((destructor)) static void pool_shutdown(void) {
    pthread_mutex_lock(&g_pool.busy);
    pthread_mutex_lock(&g_pool.lock);
    const int n = g_pool.n;
    g_pool.stop = 1;
    g_pool.n = 0;
    pthread_cond_broadcast(&g_pool.go);
    pthread_mutex_unlock(&g_pool.lock);
    for (int i = 0; i < n; i++) pthread_join(g_pool.th[i], NULL);
    pthread_mutex_unlock(&g_pool.busy);
}

static void pool_init(void) {
    const char *e = getenv("SVT_HIP_COPY_THREADS");
    int         n = e ? atoi(e) : 4;
    if (n < 1) n = 1;
    if (n > POOL_MAX + 1) n = POOL_MAX + 1;
    int made = 0;
    for (int i = 0; i < n - 1; i++) {
        if (pthread_create(&g_pool.th[made], NULL, worker, (void *)(intptr_t)made) == 0) made++;
        if (made != i + 1) break; /* no more threads to be had: the pool stays as big as it got */
    }
    g_pool.n = made;
    pthread_atfork(NULL, NULL, pool_atfork_child);
}

/* copies the rows of up to 4 planes (one picture) in ONE fork-join; returns when all of them have arrived */
void svt_copy_planes_mt(int n_planes, uint8_t *const *dst, const size_t *dst_stride, const uint8_t *const *src, const size_t *src_stride, const size_t *width,
                        const size_t *rows) {
    pthread_once(&g_once, pool_init);
    size_t total = 0;
    for (int p = 0; p < n_planes; p++) total += width[p] * rows[p];
    if (n_planes > POOL_PLANES || g_pool.n == 0 || total < (1u << 20)) { /* small pictures: the hand-over costs more than it saves */
        for (int p = 0; p < n_planes; p++) copy_rows(dst[p], dst_stride[p], src[p], src_stride[p], width[p], 0, rows[p]);
        return;
    }
    pthread_mutex_lock(&g_pool.busy);
    pthread_mutex_lock(&g_pool.lock);
    g_pool.n_planes = n_planes;
    for (int p = 0; p < n_planes; p++) {
        g_pool.dst[p] = dst[p]; g_pool.src[p] = src[p]; g_pool.dst_stride[p] = dst_stride[p]; g_pool.src_stride[p] = src_stride[p];
        g_pool.width[p] = width[p]; g_pool.rows[p] = rows[p];
    }
    g_pool.pending = g_pool.n;
    g_pool.gen++;
    pthread_cond_broadcast(&g_pool.go);
    pthread_mutex_unlock(&g_pool.lock);
    copy_slice(0, g_pool.n + 1);
    pthread_mutex_lock(&g_pool.lock);
    while (g_pool.pending) pthread_cond_wait(&g_pool.done, &g_pool.lock);
    pthread_mutex_unlock(&g_pool.lock);
    pthread_mutex_unlock(&g_pool.busy);
}

/* copies `rows` rows of `width` bytes; returns when all of them have arrived */
void svt_copy_rows_mt(uint8_t *dst, size_t dst_stride, const uint8_t *src, size_t src_stride, size_t width, size_t rows) {
    svt_copy_planes_mt(1, &dst, &dst_stride, &src, &src_stride, &width, &rows);
}
