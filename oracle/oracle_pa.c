/*
 * oracle_pa.c -- CPU restatement (TEST INFRASTRUCTURE ONLY) of the picture-analysis pre-ME stage: copy + border
 * replication of the input luma and the 1/4, 1/16 point-decimated planes.
 *   decimate_input_picture        Source/Lib/Codec/EbPictureAnalysisProcess.c:5025-5088
 *   eb_vp9_decimation_2d          Source/Lib/Codec/EbPictureAnalysisProcess.c:102-122
 *   eb_vp9_generate_padding       Source/Lib/Codec/EbMcp.c:17-58 (horizontal replication of every row, then the padded
 *                                 first / last row copied upwards / downwards)
 */
#include <stddef.h>
#include <string.h>
#include "oracle.h"

static void decimation_2d(const uint8_t *in, int in_stride, int w, int h, uint8_t *out, int out_stride, int step) {
    for (int y = 0; y < h; y += step) {
        for (int x = 0; x < w; x += step) out[x / step] = in[x];
        in += (size_t)in_stride * step;
        out += out_stride;
    }
}
static void generate_padding(uint8_t *pic, int stride, int w, int h, int pad_w, int pad_h) {
    uint8_t *row = pic + pad_w + (size_t)pad_h * stride;
    for (int y = 0; y < h; y++, row += stride) {
        memset(row - pad_w, row[0], (size_t)pad_w);
        memset(row + w, row[w - 1], (size_t)pad_w);
    }
    uint8_t *top = pic + (size_t)pad_h * stride, *bot = pic + (size_t)(pad_h + h - 1) * stride;
    for (int y = 1; y <= pad_h; y++) {
        memcpy(top - (size_t)y * stride, top, (size_t)stride); /* the reference copies `stride` bytes of the padded row */
        memcpy(bot + (size_t)y * stride, bot, (size_t)stride);
    }
}

int32_t svt_oracle_pa_prepare(const uint8_t *luma, int32_t luma_stride, const svt_pa_picture *out, int32_t make_quarter) {
    const svt_plane *pl[3] = {&out->full, &out->quarter, &out->sixteenth};
    const int        W = out->full.width, H = out->full.height;
    for (int s = 0; s < 3; s++) {
        if (s == 1 && !make_quarter) continue;
        const int step = 1 << s;
        uint8_t  *buf = (uint8_t *)pl[s]->buf;
        decimation_2d(luma, luma_stride, W, H, buf + pl[s]->origin_x + (size_t)pl[s]->origin_y * pl[s]->stride, pl[s]->stride, step);
        generate_padding(buf, pl[s]->stride, W / step, H / step, pl[s]->origin_x, pl[s]->origin_y);
    }
    return 0;
}

/* pad_ref_and_set_flags (Source/Lib/Codec/EbEncDecProcess.c:4822-4851): the deblocked reconstruction of a reference picture is
 * padded in place, Y with (origin_x, origin_y), Cb and Cr with half of it (width >> 1, height >> 1, origin >> 1) */
int32_t svt_oracle_ref_pad(const svt_yuv_planes *pic, int32_t pad_x, int32_t pad_y) {
    generate_padding(pic->y - pad_x - (ptrdiff_t)pad_y * pic->y_stride, pic->y_stride, pic->width, pic->height, pad_x, pad_y);
    generate_padding(pic->u - (pad_x >> 1) - (ptrdiff_t)(pad_y >> 1) * pic->uv_stride, pic->uv_stride, pic->width >> 1, pic->height >> 1, pad_x >> 1, pad_y >> 1);
    generate_padding(pic->v - (pad_x >> 1) - (ptrdiff_t)(pad_y >> 1) * pic->uv_stride, pic->uv_stride, pic->width >> 1, pic->height >> 1, pad_x >> 1, pad_y >> 1);
    return 0;
}

/* compute_block_mean_compute_variance (Codec/EbPictureAnalysisProcess.c:2115-3356), the path taken without AVX2:
 * eb_vp9_compute_sub_mean8x8_sse2_intrin / eb_vp9_compute_subd_mean_of_squared_values8x8_sse2_intrin
 * (ASM_SSE2/EbComputeMean_Intrinsic_SSE2.c:10-53): rows 0,2,4,6 of each 8x8; sum << 3 and sum of squares << 11 */
void svt_oracle_pa_mean8x8(const uint8_t *p, int32_t stride, uint64_t *mean, uint64_t *mean_sq) {
    uint32_t s = 0, q = 0;
    for (int r = 0; r < 8; r += 2)
        for (int x = 0; x < 8; x++) { s += p[r * stride + x]; q += (uint32_t)p[r * stride + x] * p[r * stride + x]; }
    *mean = (uint64_t)s << 3;
    *mean_sq = (uint64_t)q << 11;
}

int32_t svt_oracle_pa_mean_variance(const svt_plane *full, uint8_t *mean_out, uint16_t *var_out) {
    const int nx = (full->width + 63) / 64, ny = (full->height + 63) / 64;
    for (int sb = 0; sb < nx * ny; sb++) {
        uint64_t m[85], q[85];
        const uint8_t *o = full->buf + (size_t)(full->origin_y + (sb / nx) * 64) * full->stride + full->origin_x + (sb % nx) * 64;
        for (int b = 0; b < 64; b++) svt_oracle_pa_mean8x8(o + (size_t)(b >> 3) * 8 * full->stride + (b & 7) * 8, full->stride, &m[21 + b], &q[21 + b]);
        for (int k = 0; k < 16; k++) {
            const int b = 21 + (k >> 2) * 16 + (k & 3) * 2;
            m[5 + k] = (m[b] + m[b + 1] + m[b + 8] + m[b + 9]) >> 2;
            q[5 + k] = (q[b] + q[b + 1] + q[b + 8] + q[b + 9]) >> 2;
        }
        for (int k = 0; k < 4; k++) {
            const int b = 5 + (k >> 1) * 8 + (k & 1) * 2;
            m[1 + k] = (m[b] + m[b + 1] + m[b + 4] + m[b + 5]) >> 2;
            q[1 + k] = (q[b] + q[b + 1] + q[b + 4] + q[b + 5]) >> 2;
        }
        m[0] = (m[1] + m[2] + m[3] + m[4]) >> 2;
        q[0] = (q[1] + q[2] + q[3] + q[4]) >> 2;
        for (int i = 0; i < 85; i++) {
            mean_out[(size_t)sb * 85 + i] = (uint8_t)(m[i] >> 8);
            var_out[(size_t)sb * 85 + i]  = (uint16_t)((q[i] - m[i] * m[i]) >> 16);
        }
    }
    return 0;
}
