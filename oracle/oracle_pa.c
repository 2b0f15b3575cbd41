/*
 * oracle_pa.c -- CPU restatement (TEST INFRASTRUCTURE ONLY) of the picture-analysis pre-ME stage: copy + border
 * replication of the input luma and the 1/4, 1/16 point-decimated planes.
 *   decimate_input_picture        Source/Lib/Codec/EbPictureAnalysisProcess.c:5025-5088
 *   eb_vp9_decimation_2d          Source/Lib/Codec/EbPictureAnalysisProcess.c:102-122
 *   eb_vp9_generate_padding       Source/Lib/Codec/EbMcp.c:17-58 (horizontal replication of every row, then the padded
 *                                 first / last row copied upwards / downwards)
 */
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
