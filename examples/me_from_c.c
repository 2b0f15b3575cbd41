/*
 * me_from_c.c -- the C ABI used from plain C, the way the reference's ME kernel thread would call it (INTEGRATION.md
 * section 1): build the three padded planes of two pictures on the host, derive the ME parameters of a BASELINE
 * configuration, run motion estimation of one picture against one reference on the GPU, print a checksum of the
 * MeCuResults-compatible records and the motion found.  Exits non-zero if there is no GPU: the library has no CPU path.
 *
 *   gcc -std=c11 -O2 -Iinclude examples/me_from_c.c -Lsvt-vp9_amd -lsvtvp9_hip -Wl,-rpath,$PWD/svt-vp9_amd -o me_from_c
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "svtvp9_hip.h"

/* decimate by `step` and replicate the borders by `pad` samples: what picture analysis hands to ME
 * (Codec/EbPictureAnalysisProcess.c:102-122, 5010-5088) */
static svt_plane make_plane(const uint8_t *luma, int w, int h, int step, int pad) {
    const int pw = w / step, ph = h / step, stride = pw + 2 * pad;
    uint8_t  *buf = (uint8_t *)malloc((size_t)stride * (ph + 2 * pad));
    for (int y = -pad; y < ph + pad; y++)
        for (int x = -pad; x < pw + pad; x++) {
            const int sy = (y < 0 ? 0 : y >= ph ? ph - 1 : y) * step, sx = (x < 0 ? 0 : x >= pw ? pw - 1 : x) * step;
            buf[(size_t)(y + pad) * stride + x + pad] = luma[(size_t)sy * w + sx];
        }
    svt_plane p = {buf, stride, pad, pad, pw, ph};
    return p;
}
static svt_pa_picture make_picture(const uint8_t *luma, int w, int h) {
    svt_pa_picture p = {make_plane(luma, w, h, 1, 68), make_plane(luma, w, h, 2, 32), make_plane(luma, w, h, 4, 16)};
    return p;
}

int main(void) {
    const int W = 640, H = 384, DX = 5, DY = -3; /* the current picture is the reference moved by (DX, DY) samples */
    uint8_t  *ref = (uint8_t *)malloc((size_t)W * H), *cur = (uint8_t *)malloc((size_t)W * H);
    uint8_t  *noise = (uint8_t *)malloc((size_t)(W + 4) * (H + 4));
    uint32_t  s = 12345;
    for (int i = 0; i < (W + 4) * (H + 4); i++) { s = s * 1664525u + 1013904223u; noise[i] = (uint8_t)(s >> 24); }
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) { /* low-passed noise: aperiodic, so the displaced copy matches at one position only */
            int a = 0;
            for (int j = 0; j < 4; j++)
                for (int i = 0; i < 4; i++) a += noise[(y + j) * (W + 4) + x + i];
            ref[y * W + x] = (uint8_t)(a >> 4);
        }
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const int sy = y + DY < 0 ? 0 : y + DY >= H ? H - 1 : y + DY, sx = x + DX < 0 ? 0 : x + DX >= W ? W - 1 : x + DX;
            cur[y * W + x] = ref[sy * W + sx];
        }
    svt_pa_picture pc = make_picture(cur, W, H), pr = make_picture(ref, W, H);

    svt_me_params prm;
    if (svt_hip_me_params_preset(&prm, 1920, 1080, 8, 1, /*num_ref_lists=*/1, /*temporal_layer=*/0, /*hierarchical_levels=*/4) != SVT_HIP_OK) {
        fprintf(stderr, "preset: %s\n", svt_hip_last_error());
        return 2;
    }
    svt_hip_ctx *ctx = NULL;
    if (svt_hip_ctx_create(&ctx, 0) != SVT_HIP_OK) {
        fprintf(stderr, "no usable GPU: %s\n", svt_hip_last_error());
        return 3;
    }
    const int         n_sb = svt_hip_sb_count(W, H);
    svt_me_pu_result *res = (svt_me_pu_result *)calloc((size_t)n_sb * 85, sizeof *res);
    if (svt_hip_me_picture(ctx, &pc, &pr, NULL, &prm, res, NULL) != SVT_HIP_OK) {
        fprintf(stderr, "me: %s\n", svt_hip_last_error());
        return 4;
    }
    uint32_t crc = 0;
    int      hits = 0;
    for (int i = 0; i < n_sb * 85; i++) {
        const uint32_t *w = (const uint32_t *)&res[i];
        for (int k = 0; k < 8; k++) crc = (crc << 5 | crc >> 27) ^ w[k];
        if (res[i].x_mv_l0 == 4 * DX && res[i].y_mv_l0 == 4 * DY) hits++; /* quarter-sample units */
    }
    printf("superblocks %d  checksum %08x  partitions on the true motion %d / %d  (%.3f ms on the GPU)\n", n_sb, crc, hits, n_sb * 85,
           svt_hip_last_kernel_ms(ctx));
    svt_hip_ctx_destroy(ctx);
    return hits > n_sb * 85 / 2 ? 0 : 5;
}
