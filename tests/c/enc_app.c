/*
 * enc_app.c -- a caller written against the SVT-VP9 public API the way the reference's sample application drives it
 * (App/EbAppContext.c:355-427: init_handle -> fill the configuration -> set_parameter -> init_encoder -> stream_header;
 * App/EbAppProcessCmd.c:437-683: send_picture per frame with the EOS flag on the last, get_packet / release_out_buffer until the
 * EOS packet; then deinit_encoder -> deinit_handle).  It includes whichever header SVT_API_HEADER names (this repository's
 * or the reference's: the test compiles it against both) and links libSvtVp9Enc.
 *
 *   enc_app in.yuv W H frames enc_mode tune   -> prints "packets N eos E bytes B"; exit 3 = no GPU (init_encoder refused)
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include SVT_API_HEADER

int main(int argc, char **argv) {
    if (argc < 7) return 2;
    const int W = atoi(argv[2]), H = atoi(argv[3]), N = atoi(argv[4]);
    EbComponentType         *h = NULL;
    EbSvtVp9EncConfiguration cfg;
    memset(&cfg, 0, sizeof cfg);
    if (eb_vp9_svt_init_handle(&h, (void *)argv, &cfg) != EB_ErrorNone || !h) return 4;
    if (cfg.enc_mode != 3 || cfg.qp != 50 || cfg.intra_period != 31) return 5; /* the library loaded its defaults */
    cfg.source_width = (uint32_t)W; cfg.source_height = (uint32_t)H;
    cfg.enc_mode = (uint8_t)atoi(argv[5]); cfg.tune = (uint8_t)atoi(argv[6]);
    cfg.frame_rate = 60 << 16; cfg.qp = 40; cfg.intra_period = -1; cfg.frames_to_be_encoded = (uint64_t)N;
    if (eb_vp9_svt_enc_set_parameter(h, &cfg) != EB_ErrorNone) return 6;
    EbErrorType e = eb_vp9_init_encoder(h);
    if (e == EB_ErrorInsufficientResources) { printf("no device\n"); eb_vp9_deinit_handle(h); return 3; }
    if (e != EB_ErrorNone) return 7;
    EbBufferHeaderType *hdr = NULL;
    if (eb_vp9_svt_enc_stream_header(h, &hdr) != EB_ErrorNone) return 8;

    FILE *f = fopen(argv[1], "rb");
    if (!f) return 9;
    const size_t ysz = (size_t)W * H, csz = ysz / 4;
    uint8_t     *buf = (uint8_t *)malloc(ysz + 2 * csz);
    int          packets = 0, eos = 0;
    long long    bytes = 0;
    for (int n = 0; n < N; n++) {
        if (fread(buf, 1, ysz + 2 * csz, f) != ysz + 2 * csz) return 10;
        EbSvtEncInput in;
        memset(&in, 0, sizeof in);
        in.luma = buf; in.cb = buf + ysz; in.cr = buf + ysz + csz;
        in.y_stride = (uint32_t)W; in.cb_stride = in.cr_stride = (uint32_t)W / 2;
        EbBufferHeaderType b;
        memset(&b, 0, sizeof b);
        b.size = sizeof b; b.p_buffer = (uint8_t *)&in; b.n_filled_len = (uint32_t)(ysz + 2 * csz); b.pts = n;
        b.flags = n == N - 1 ? EB_BUFFERFLAG_EOS : 0;
        if (eb_vp9_svt_enc_send_picture(h, &b) != EB_ErrorNone) return 11;
        memset(buf, 0xEE, ysz + 2 * csz); /* the library has copied the picture: the caller's buffer is its own again */
        for (;;) { /* drain what is ready (non-blocking poll while pictures are still being sent) */
            EbBufferHeaderType *p = NULL;
            e = eb_vp9_svt_get_packet(h, &p, (uint8_t)(n == N - 1));
            if (e == EB_NoErrorEmptyQueue) break;
            if (e != EB_ErrorNone || !p) return 12;
            packets++; bytes += p->n_filled_len; eos |= (p->flags & EB_BUFFERFLAG_EOS) != 0;
            eb_vp9_svt_release_out_buffer(&p);
            if (p) return 13;
        }
    }
    fclose(f);
    free(buf);
    EbBufferHeaderType rec;
    memset(&rec, 0, sizeof rec);
    if (eb_vp9_svt_get_recon(h, &rec) != EB_ErrorMax) return 14; /* recon_file = 0 */
    if (eb_vp9_deinit_encoder(h) != EB_ErrorNone) return 15;
    if (eb_vp9_deinit_handle(h) != EB_ErrorNone) return 16;
    printf("packets %d eos %d bytes %lld\n", packets, eos, bytes);
    return 0;
}
