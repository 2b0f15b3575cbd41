/*
 * svt_hip_me_app.c -- thin command-line front end over the first stage of the path (scaled to what this
 * repository owns): reads 8-bit 4:2:0 planar pictures the way the reference's sample application does (App/EbAppProcessCmd.c
 * read_input_frames: Y then U then V per picture, no header), derives the three ME planes of every picture, runs motion
 * estimation of picture k against picture k-1 (list 0) and -- with -b -- picture k+1 (list 1) through the C ABI, and writes the
 * MeCuResults-compatible records.  With -ivf the record stream of every picture is wrapped as one IVF frame by the library's
 * container layer (the container is the reference application's; the payload is ME records, not a VP9 bitstream: the
 * entropy coder is outside this repository's scope).  Plain C11; exits 3 when there is no GPU -- the library has no CPU path.
 *
 *   gcc -std=c11 -O2 -Iinclude app/svt_hip_me_app.c -Lsvt-vp9_amd -lsvtvp9_hip -Wl,-rpath,$PWD/svt-vp9_amd -o svt_hip_me_app
 *   ./svt_hip_me_app -i in.yuv -w 1920 -h 1080 -n 30 -enc-mode 8 [-b] [-o me.bin] [-ivf me.ivf] [-fps 60]
 */
#define _POSIX_C_SOURCE 200809L /* clock_gettime */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "svtvp9_hip.h"

static svt_plane make_plane(const uint8_t *luma, int w, int h, int step, int pad) {
    const int pw = w / step, ph = h / step, stride = pw + 2 * pad;
    uint8_t  *buf = (uint8_t *)malloc((size_t)stride * (ph + 2 * pad));
    for (int y = -pad; y < ph + pad; y++) {
        const uint8_t *srow = luma + (size_t)((y < 0 ? 0 : y >= ph ? ph - 1 : y) * step) * w;
        uint8_t       *drow = buf + (size_t)(y + pad) * stride + pad;
        for (int x = 0; x < pw; x++) drow[x] = srow[x * step];
        memset(drow - pad, drow[0], (size_t)pad);
        memset(drow + pw, drow[pw - 1], (size_t)pad);
    }
    svt_plane p = {buf, stride, pad, pad, pw, ph};
    return p;
}
/* full, 1/4 and 1/16 resolution with the reference's paddings (Codec/EbPictureAnalysisProcess.c:102-122, 5010-5088) */
static svt_pa_picture make_picture(const uint8_t *luma, int w, int h) {
    svt_pa_picture p = {make_plane(luma, w, h, 1, 68), make_plane(luma, w, h, 2, 32), make_plane(luma, w, h, 4, 16)};
    return p;
}
static void free_picture(svt_pa_picture *p) {
    free((void *)p->full.buf); free((void *)p->quarter.buf); free((void *)p->sixteenth.buf);
    memset(p, 0, sizeof *p);
}
static int usage(void) {
    fprintf(stderr, "usage: svt_hip_me_app -i in.yuv -w W -h H [-n frames] [-enc-mode 0..9] [-b] [-o records.bin] [-ivf records.ivf] [-fps F]\n");
    return 2;
}

int main(int argc, char **argv) {
    const char *in = NULL, *out = NULL, *ivf = NULL;
    int         W = 0, H = 0, n = 0, enc_mode = 8, bipred = 0, fps = 60;
    for (int i = 1; i < argc; i++) {
        const char *a = argv[i], *v = i + 1 < argc ? argv[i + 1] : NULL;
        if (!strcmp(a, "-b")) bipred = 1;
        else if (!v) return usage();
        else if (!strcmp(a, "-i")) { in = v; i++; }
        else if (!strcmp(a, "-o")) { out = v; i++; }
        else if (!strcmp(a, "-ivf")) { ivf = v; i++; }
        else if (!strcmp(a, "-w")) { W = atoi(v); i++; }
        else if (!strcmp(a, "-h")) { H = atoi(v); i++; }
        else if (!strcmp(a, "-n")) { n = atoi(v); i++; }
        else if (!strcmp(a, "-fps")) { fps = atoi(v); i++; }
        else if (!strcmp(a, "-enc-mode")) { enc_mode = atoi(v); i++; }
        else return usage();
    }
    /* the reference accepts 64..8192 x 64..4320, multiples of 8 (Codec/EbEncHandle.c:2295-2340) */
    if (!in || W < 64 || H < 64 || W > 8192 || H > 4320 || (W & 7) || (H & 7)) return usage();
    FILE *fi = fopen(in, "rb");
    if (!fi) { fprintf(stderr, "cannot open %s\n", in); return 2; }
    const size_t ysz = (size_t)W * H, fsz = ysz + ysz / 2;
    fseek(fi, 0, SEEK_END);
    const long total = ftell(fi);
    fseek(fi, 0, SEEK_SET);
    const int avail = (int)((size_t)total / fsz);
    if (n <= 0 || n > avail) n = avail;
    if (n < 2) { fprintf(stderr, "%s holds %d pictures of %dx%d: at least 2 are needed\n", in, avail, W, H); return 2; }

    svt_hip_ctx *ctx = NULL;
    if (svt_hip_ctx_create(&ctx, 0) != SVT_HIP_OK) { fprintf(stderr, "no usable GPU: %s\n", svt_hip_last_error()); return 3; }
    svt_me_params prm1, prm2;
    if (svt_hip_me_params_preset(&prm1, W, H, enc_mode, 1, 1, 0, 4) != SVT_HIP_OK || svt_hip_me_params_preset(&prm2, W, H, enc_mode, 1, 2, 1, 4) != SVT_HIP_OK) {
        fprintf(stderr, "preset: %s\n", svt_hip_last_error());
        return 2;
    }
    FILE *fo = out ? fopen(out, "wb") : NULL, *fv = ivf ? fopen(ivf, "wb") : NULL;
    if ((out && !fo) || (ivf && !fv)) { fprintf(stderr, "cannot open an output file\n"); return 2; }
    const int         n_sb = svt_hip_sb_count(W, H);
    const size_t      rbytes = (size_t)n_sb * 85 * sizeof(svt_me_pu_result);
    svt_me_pu_result *res = (svt_me_pu_result *)malloc(rbytes);
    uint8_t          *luma = (uint8_t *)malloc(fsz), *pkt = (uint8_t *)malloc(rbytes + 12);
    if (fv) {
        uint8_t h[SVT_IVF_STREAM_HEADER_BYTES];
        svt_ivf_stream_header(h, (uint32_t)W, (uint32_t)H, (uint32_t)fps << 16, 0, 0);
        memcpy(h + 8, "SVME", 4); /* the payload is ME records, not a VP9 bitstream: not the reference's "VP90" fourcc */
        fwrite(h, 1, sizeof h, fv);
    }
    /* a window of three pictures: previous, current, next */
    svt_pa_picture pic[3];
    memset(pic, 0, sizeof pic);
    for (int k = 0; k < 2; k++) {
        if (fread(luma, 1, fsz, fi) != fsz) return 2;
        pic[k + 1] = make_picture(luma, W, H);
    }
    double   gpu_ms = 0;
    uint32_t crc = 0;
    long     moved = 0;
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int k = 1; k < n; k++) { /* picture k against k-1 (and k+1) */
        free_picture(&pic[0]);
        pic[0] = pic[1]; pic[1] = pic[2]; memset(&pic[2], 0, sizeof pic[2]);
        int have_next = 0;
        if (k + 1 < n && fread(luma, 1, fsz, fi) == fsz) { pic[2] = make_picture(luma, W, H); have_next = 1; }
        const int two = bipred && have_next;
        if (svt_hip_me_picture(ctx, &pic[1], &pic[0], two ? &pic[2] : NULL, two ? &prm2 : &prm1, res, NULL) != SVT_HIP_OK) {
            fprintf(stderr, "me: %s\n", svt_hip_last_error());
            return 4;
        }
        gpu_ms += svt_hip_last_kernel_ms(ctx);
        for (size_t i = 0; i < (size_t)n_sb * 85; i++) {
            const uint32_t *w = (const uint32_t *)&res[i];
            for (size_t j = 0; j < sizeof(svt_me_pu_result) / 4; j++) crc = (crc << 5 | crc >> 27) ^ w[j];
            moved += res[i].x_mv_l0 != 0 || res[i].y_mv_l0 != 0;
        }
        if (fo) fwrite(res, 1, rbytes, fo);
        if (fv) {
            const int64_t m = svt_ivf_packetize((const uint8_t *)res, (uint32_t)rbytes, (uint64_t)k, 0, pkt, rbytes + 12);
            if (m < 0) return 4;
            fwrite(pkt, 1, (size_t)m, fv);
        }
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    const double wall = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    printf("%d pictures %dx%d enc-mode %d%s: %d superblocks, checksum %08x, %.1f %% of the partitions moved, ME kernel %.3f ms/picture, "
           "%.2f pictures/s end to end (host plane construction and PCIe included)\n",
           n - 1, W, H, enc_mode, bipred ? " two lists" : "", n_sb, crc, 100.0 * (double)moved / ((double)(n - 1) * n_sb * 85), gpu_ms / (n - 1),
           (n - 1) / wall);
    if (fo) fclose(fo);
    if (fv) fclose(fv);
    fclose(fi);
    svt_hip_ctx_destroy(ctx);
    return 0;
}
