/*
 * svt_enc_api_bench.c -- SURVEY 8(d)'s metric through the public API: wall-clock from the first eb_vp9_svt_enc_send_picture to
 * the EOS packet, init excluded, host buffers handed over (PCIe inside the clock).  Plain C against include/svt_vp9_enc_api.h
 * (the reference's EbSvtVp9Enc.h works as well), the call sequence of the reference's sample application
 * (App/EbAppProcessCmd.c:437-683): send a picture, poll get_packet without blocking, after the last picture drain with
 * pic_send_done = 1.
 *
 *   svt_enc_api_bench clip.yuv W H frames_in_file frames_to_send enc_mode tune [recon [devices]]
 *       clip.yuv holds frames_in_file 4:2:0 pictures of W x H (Y, Cb, Cr); they are sent round-robin.  All three planes cross
 *       PCIe: behind the API run picture analysis, motion estimation, the stand-in decision, inter prediction, transform /
 *       quantisation / reconstruction, deblocking and reference padding (which of the last three a picture gets is the
 *       reference's rule: svt_hip_encdec_flags_derive).  recon = 1: recon_file is set and every reconstructed picture is fetched
 *       with eb_vp9_svt_get_recon (12.4 MB per 4K picture back over PCIe), as the reference's application does with -o.
 *   prints one line: {"frames": N, "seconds": T, "frames_per_s": F, "packets": P, "me_launches": L, "recon": R, "recon_pictures": K, "drain_seconds": D}
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
    if (argc < 8) { fprintf(stderr, "usage: %s clip.yuv W H frames_in_file frames_to_send enc_mode tune [recon [devices]]\n", argv[0]); return 2; }
    /* devices: a comma-separated list of GPU ordinals the library deals the closed GOPs to (its SVT_HIP_DEVICES; an ordinal may repeat:
       several contexts, each with its own picture ring, streams and feeder thread, on one GPU) */
    if (argc > 9 && argv[9][0]) setenv("SVT_HIP_DEVICES", argv[9], 1);
    const int W = atoi(argv[2]), H = atoi(argv[3]), K = atoi(argv[4]), N = atoi(argv[5]), want_recon = argc > 8 ? atoi(argv[8]) : 0;
    if (W < 64 || H < 64 || K < 1 || N < 1) return 2;
    const size_t ysz = (size_t)W * H, psz = ysz + ysz / 2;
    uint8_t     *clip = (uint8_t *)malloc(psz * (size_t)K), *rbuf = (uint8_t *)malloc(psz);
    FILE        *f = fopen(argv[1], "rb");
    if (!clip || !rbuf || !f || fread(clip, 1, psz * (size_t)K, f) != psz * (size_t)K) { fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
    fclose(f);

    EbComponentType         *h = NULL;
    EbSvtVp9EncConfiguration cfg;
    memset(&cfg, 0, sizeof cfg);
    if (eb_vp9_svt_init_handle(&h, NULL, &cfg) != EB_ErrorNone || !h) return 4;
    cfg.source_width = (uint32_t)W; cfg.source_height = (uint32_t)H;
    cfg.enc_mode = (uint8_t)atoi(argv[6]); cfg.tune = (uint8_t)atoi(argv[7]);
    cfg.frame_rate = 60 << 16; cfg.qp = 40; cfg.intra_period = -2; cfg.frames_to_be_encoded = (uint64_t)N; cfg.recon_file = (uint32_t)want_recon;
    if (eb_vp9_svt_enc_set_parameter(h, &cfg) != EB_ErrorNone) return 6;
    const EbErrorType ie = eb_vp9_init_encoder(h);
    if (ie == EB_ErrorInsufficientResources) { printf("no device\n"); eb_vp9_deinit_handle(h); return 3; }
    if (ie != EB_ErrorNone) return 7;

    int          packets = 0, eos = 0, recons = 0, recon_eos = 0;
    double       t_last_sent = 0.0; /* when the last send_picture returned: what follows is the drain (the last group's ME + its layers) */
    const double t0 = now_s();
    for (int n = 0; n < N; n++) {
        EbSvtEncInput in;
        memset(&in, 0, sizeof in);
        in.luma = clip + psz * (size_t)(n % K); in.cb = in.luma + ysz; in.cr = in.cb + ysz / 4;
        in.y_stride = (uint32_t)W; in.cb_stride = in.cr_stride = (uint32_t)W / 2;
        EbBufferHeaderType b;
        memset(&b, 0, sizeof b);
        b.size = sizeof b; b.p_buffer = (uint8_t *)&in; b.n_filled_len = (uint32_t)(ysz + ysz / 2); b.pts = n;
        b.flags = n == N - 1 ? EB_BUFFERFLAG_EOS : 0;
        if (eb_vp9_svt_enc_send_picture(h, &b) != EB_ErrorNone) return 11;
        if (n == N - 1) t_last_sent = now_s();
        for (;;) {
            EbBufferHeaderType *p = NULL;
            const EbErrorType   e = eb_vp9_svt_get_packet(h, &p, (uint8_t)(n == N - 1));
            if (e == EB_NoErrorEmptyQueue) break;
            if (e != EB_ErrorNone || !p) return 12;
            packets++;
            eos |= (p->flags & EB_BUFFERFLAG_EOS) != 0;
            eb_vp9_svt_release_out_buffer(&p);
        }
        while (want_recon && !recon_eos) { /* what is ready of the reconstructed pictures; after the last picture: all of them */
            EbBufferHeaderType r;
            memset(&r, 0, sizeof r);
            r.size = sizeof r; r.p_buffer = rbuf; r.n_alloc_len = (uint32_t)psz;
            const EbErrorType e = eb_vp9_svt_get_recon(h, &r);
            if (e == EB_NoErrorEmptyQueue) { if (n == N - 1) continue; break; }
            if (e != EB_ErrorNone) return 14;
            recons++;
            recon_eos = (r.flags & EB_BUFFERFLAG_EOS) != 0;
        }
    }
    const double t1 = now_s();
    uint64_t     launches = 0, sent = 0;
    (void)svt_vp9_shim_get_counters(h, &launches, &sent);
    if (!eos || packets != N || (want_recon && recons != N)) { fprintf(stderr, "packets %d of %d, eos %d, reconstructions %d\n", packets, N, eos, recons); return 13; }
    printf("{\"frames\": %d, \"seconds\": %.6f, \"frames_per_s\": %.2f, \"packets\": %d, \"me_launches\": %llu, \"recon\": %d, \"recon_pictures\": %d, \"drain_seconds\": %.6f}\n", N,
           t1 - t0, N / (t1 - t0), packets, (unsigned long long)launches, want_recon, recons, t1 - t_last_sent);
    eb_vp9_deinit_encoder(h);
    eb_vp9_deinit_handle(h);
    free(clip);
    free(rbuf);
    return 0;
}
