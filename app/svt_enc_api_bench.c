/*
 * svt_enc_api_bench.c -- SURVEY 8(d)'s metric through the public API: wall-clock from the first eb_vp9_svt_enc_send_picture to
 * the EOS packet, init excluded, host buffers handed over (PCIe inside the clock).  Plain C against include/svt_vp9_enc_api.h
 * (the reference's EbSvtVp9Enc.h works as well), the call sequence of the reference's sample application
 * (App/EbAppProcessCmd.c:437-683): send a picture, poll get_packet without blocking, after the last picture drain with
 * pic_send_done = 1.
 *
 *   svt_enc_api_bench luma.bin W H frames_in_file frames_to_send enc_mode tune
 *       luma.bin holds frames_in_file luma planes of W x H (the library reads luma only: picture analysis and motion
 *       estimation are what runs behind the API); they are sent round-robin.
 *   prints one line: {"frames": N, "seconds": T, "frames_per_s": F, "packets": P, "me_launches": L}
 *   exit 3 = no GPU (init_encoder refused: the library has no CPU path)
 */
#define _POSIX_C_SOURCE 200809L
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "svt_vp9_enc_api.h"

static double now_s(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

int main(int argc, char **argv) {
    if (argc < 8) { fprintf(stderr, "usage: %s luma.bin W H frames_in_file frames_to_send enc_mode tune\n", argv[0]); return 2; }
    const int W = atoi(argv[2]), H = atoi(argv[3]), K = atoi(argv[4]), N = atoi(argv[5]);
    if (W < 64 || H < 64 || K < 1 || N < 1) return 2;
    const size_t ysz = (size_t)W * H;
    uint8_t     *clip = (uint8_t *)malloc(ysz * (size_t)K), *chroma = (uint8_t *)malloc(ysz / 4);
    FILE        *f = fopen(argv[1], "rb");
    if (!clip || !chroma || !f || fread(clip, 1, ysz * (size_t)K, f) != ysz * (size_t)K) { fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
    fclose(f);
    memset(chroma, 128, ysz / 4);

    EbComponentType         *h = NULL;
    EbSvtVp9EncConfiguration cfg;
    memset(&cfg, 0, sizeof cfg);
    if (eb_vp9_svt_init_handle(&h, NULL, &cfg) != EB_ErrorNone || !h) return 4;
    cfg.source_width = (uint32_t)W; cfg.source_height = (uint32_t)H;
    cfg.enc_mode = (uint8_t)atoi(argv[6]); cfg.tune = (uint8_t)atoi(argv[7]);
    cfg.frame_rate = 60 << 16; cfg.qp = 40; cfg.intra_period = -2; cfg.frames_to_be_encoded = (uint64_t)N;
    if (eb_vp9_svt_enc_set_parameter(h, &cfg) != EB_ErrorNone) return 6;
    const EbErrorType ie = eb_vp9_init_encoder(h);
    if (ie == EB_ErrorInsufficientResources) { printf("no device\n"); eb_vp9_deinit_handle(h); return 3; }
    if (ie != EB_ErrorNone) return 7;

    int          packets = 0, eos = 0;
    const double t0 = now_s();
    for (int n = 0; n < N; n++) {
        EbSvtEncInput in;
        memset(&in, 0, sizeof in);
        in.luma = clip + ysz * (size_t)(n % K); in.cb = chroma; in.cr = chroma;
        in.y_stride = (uint32_t)W; in.cb_stride = in.cr_stride = (uint32_t)W / 2;
        EbBufferHeaderType b;
        memset(&b, 0, sizeof b);
        b.size = sizeof b; b.p_buffer = (uint8_t *)&in; b.n_filled_len = (uint32_t)(ysz + ysz / 2); b.pts = n;
        b.flags = n == N - 1 ? EB_BUFFERFLAG_EOS : 0;
        if (eb_vp9_svt_enc_send_picture(h, &b) != EB_ErrorNone) return 11;
        for (;;) {
            EbBufferHeaderType *p = NULL;
            const EbErrorType   e = eb_vp9_svt_get_packet(h, &p, (uint8_t)(n == N - 1));
            if (e == EB_NoErrorEmptyQueue) break;
            if (e != EB_ErrorNone || !p) return 12;
            packets++;
            eos |= (p->flags & EB_BUFFERFLAG_EOS) != 0;
            eb_vp9_svt_release_out_buffer(&p);
        }
    }
    const double t1 = now_s();
    uint64_t     launches = 0, sent = 0;
    (void)svt_vp9_shim_get_counters(h, &launches, &sent);
    if (!eos || packets != N) { fprintf(stderr, "packets %d of %d, eos %d\n", packets, N, eos); return 13; }
    printf("{\"frames\": %d, \"seconds\": %.6f, \"frames_per_s\": %.2f, \"packets\": %d, \"me_launches\": %llu}\n", N, t1 - t0, N / (t1 - t0), packets,
           (unsigned long long)launches);
    eb_vp9_deinit_encoder(h);
    eb_vp9_deinit_handle(h);
    free(clip);
    free(chroma);
    return 0;
}
