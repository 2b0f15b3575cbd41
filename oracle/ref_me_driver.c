/*
 * ref_me_driver.c -- harness that runs the REFERENCE's motion_estimate_sb
 * (/root/reference/Source/Lib/Codec/EbMotionEstimation.c:4524) on a picture described by a binary
 * request file, and writes the reference's MeCuResults.  TEST INFRASTRUCTURE ONLY.
 *
 * It is compiled only in the build container (it #includes the reference's headers from
 * /root/reference at build time; nothing of the reference is copied into this repository) and linked
 * with the reference's own objects into oracle/_ref/ref_me_sb.  This file contains no reference code:
 * it allocates the structures motion_estimate_sb reads, fills the fields from the request exactly as
 * the ME process does (Codec/EbMotionEstimationProcess.c:964-1044) and calls the reference.
 *
 * The only symbol defined here on behalf of the library is the dispatch selector eb_vp9_ASM_TYPES
 * (declared extern at Codec/EbDefinitions.h:215, normally set from the `-asm` CLI flag at
 * Codec/EbEncHandle.c:800-804): 0 selects the C_DEFAULT kernels = `-asm 0`.
 *
 * request file layout (little endian):
 *   int32 magic 'SVME', svt_me_params (raw), int32 sb_begin, int32 sb_end,
 *   9 planes (cur.full, cur.quarter, cur.sixteenth, ref0.*, ref1.*), each:
 *   int32 stride, origin_x, origin_y, width, height, nbytes, then nbytes of samples (nbytes may be 0).
 * response: int32 n_sb, then n_sb*85 svt_me_pu_result, then n_sb uint32 rcme distortion, then one double: the seconds the SB
 *           loop took (clock_gettime around the motion_estimate_sb calls only: the figure bench.py's cpu_baseline reports for
 *           the reference's own C path, free of this harness's file I/O).
 *
 * Second request kind (the one leaf of the SSD fractional search that has external linkage and runs without the yasm-only
 * Log2f: eb_vp9_combined_averaging_ssd, Codec/EbMotionEstimation.c:1708-1725, the quarter-pel metric of SSD_SEARCH):
 *   int32 magic 'SVAS', int32 n, then n jobs {int32 width, height, src_stride, ref1_stride, ref2_stride; src[height *
 *   src_stride]; ref1[height * ref1_stride]; ref2[height * ref2_stride]}
 * response: n uint32
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <time.h>

#include "EbDefinitions.h"
#include "EbPictureControlSet.h"
#include "EbSequenceControlSet.h"
#include "EbMotionEstimation.h"
#include "EbMotionEstimationContext.h"
#include "EbReferenceObject.h"
#include "EbPictureBufferDesc.h"

#include "../include/svtvp9_hip.h"

uint32_t eb_vp9_ASM_TYPES = 0; /* `-asm 0`: C_DEFAULT path */

static int rd(FILE *f, void *p, size_t n) { return fread(p, 1, n, f) == n ? 0 : -1; }

static EbPictureBufferDesc *read_plane(FILE *f) {
    int32_t              h[6];
    EbPictureBufferDesc *d = (EbPictureBufferDesc *)calloc(1, sizeof *d);
    if (rd(f, h, sizeof h)) exit(3);
    d->stride_y = (uint16_t)h[0];
    d->origin_x = (uint16_t)h[1];
    d->origin_y = (uint16_t)h[2];
    d->width    = (uint16_t)h[3];
    d->height   = (uint16_t)h[4];
    if (h[5] > 0) {
        d->buffer_y = (uint8_t *)malloc((size_t)h[5] + 4096);
        if (rd(f, d->buffer_y, (size_t)h[5])) exit(3);
    }
    return d;
}

int main(int argc, char **argv) {
    if (argc < 3) { fprintf(stderr, "usage: %s request.bin response.bin\n", argv[0]); return 2; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    int32_t       magic, sb_begin, sb_end;
    svt_me_params p;
    if (rd(f, &magic, 4)) return 3;
    if (magic == 0x53415653) { /* 'SVAS' */
        int32_t n;
        if (rd(f, &n, 4) || n < 0) return 3;
        uint32_t *out = (uint32_t *)calloc((size_t)n + 1, sizeof *out);
        for (int i = 0; i < n; i++) {
            int32_t g[5];
            if (rd(f, g, sizeof g)) return 3;
            uint8_t *b[3];
            for (int k = 0; k < 3; k++) {
                const size_t nb = (size_t)g[1] * g[2 + k];
                b[k] = (uint8_t *)malloc(nb + 64);
                if (rd(f, b[k], nb)) return 3;
            }
            out[i] = eb_vp9_combined_averaging_ssd(b[0], (uint32_t)g[2], b[1], (uint32_t)g[3], b[2], (uint32_t)g[4], (uint32_t)g[1], (uint32_t)g[0]);
            for (int k = 0; k < 3; k++) free(b[k]);
        }
        fclose(f);
        FILE *o = fopen(argv[2], "wb");
        if (!o) return 2;
        fwrite(out, sizeof *out, (size_t)n, o);
        fclose(o);
        return 0;
    }
    if (magic != 0x454D5653) return 3;
    if (rd(f, &p, sizeof p) || rd(f, &sb_begin, 4) || rd(f, &sb_end, 4)) return 3;
    EbPictureBufferDesc *pl[9];
    for (int i = 0; i < 9; i++) pl[i] = read_plane(f);
    fclose(f);

    /* sequence / picture control sets: only the fields motion_estimate_sb reads */
    SequenceControlSet *scs = (SequenceControlSet *)calloc(1, sizeof *scs);
    scs->luma_width                      = pl[0]->width;
    scs->luma_height                     = pl[0]->height;
    scs->static_config.rate_control_mode = p.rate_control_mode;
    EbObjectWrapper scs_w                = {0};
    scs_w.object_ptr                     = scs;

    EbPaReferenceObject cur_obj = {0}, ref_obj[2] = {{0}, {0}};
    cur_obj.input_padded_picture_ptr        = pl[0];
    cur_obj.quarter_decimated_picture_ptr   = pl[1];
    cur_obj.sixteenth_decimated_picture_ptr = pl[2];
    for (int l = 0; l < 2; l++) {
        ref_obj[l].input_padded_picture_ptr        = pl[3 + 3 * l];
        ref_obj[l].quarter_decimated_picture_ptr   = pl[4 + 3 * l];
        ref_obj[l].sixteenth_decimated_picture_ptr = pl[5 + 3 * l];
    }
    EbObjectWrapper ref_w[2] = {{0}, {0}};
    ref_w[0].object_ptr      = &ref_obj[0];
    ref_w[1].object_ptr      = &ref_obj[1];

    PictureParentControlSet *pcs         = (PictureParentControlSet *)calloc(1, sizeof *pcs);
    pcs->sequence_control_set_wrapper_ptr = &scs_w;
    pcs->slice_type                       = p.num_ref_lists == 2 ? B_SLICE : P_SLICE;
    pcs->temporal_layer_index             = p.temporal_layer_index;
    pcs->hierarchical_levels              = p.hierarchical_levels;
    pcs->enable_hme_flag                  = p.enable_hme_flag;
    pcs->enable_hme_level_0_flag          = p.enable_hme_level_0_flag;
    pcs->enable_hme_level_1_flag          = p.enable_hme_level_1_flag;
    pcs->enable_hme_level_2_flag          = p.enable_hme_level_2_flag;
    pcs->cu8x8_mode                       = (EbCu8x8Mode)p.cu8x8_mode;
    pcs->cu16x16_mode                     = (EbCu16x16Mode)p.cu16x16_mode;
    pcs->max_number_of_pus_per_sb         = SQUARE_PU_COUNT;
    pcs->ref_pa_pic_ptr_array[0]          = &ref_w[0];
    pcs->ref_pa_pic_ptr_array[1]          = &ref_w[1];
    pcs->ref_pic_poc_array[0]             = 8;
    pcs->ref_pic_poc_array[1]             = p.same_ref_poc ? 8 : 16;

    const int W = pl[0]->width, H = pl[0]->height;
    const int nx = (W + 63) / 64, ny = (H + 63) / 64, n_sb = nx * ny;
    if (sb_end < 0 || sb_end > n_sb) sb_end = n_sb;
    pcs->me_results = (MeCuResults **)calloc((size_t)n_sb, sizeof(MeCuResults *));
    for (int i = 0; i < n_sb; i++) pcs->me_results[i] = (MeCuResults *)calloc(SQUARE_PU_COUNT, sizeof(MeCuResults));
    pcs->rcme_distortion = (uint32_t *)calloc((size_t)n_sb, sizeof(uint32_t));

    /* MeContext: buffers sized as eb_vp9_me_context_ctor does (Codec/EbMotionEstimationContext.c) */
    MeContext *mc                  = (MeContext *)calloc(1, sizeof *mc);
    mc->sb_buffer_stride           = MAX_SB_SIZE;
    mc->sb_buffer                  = (uint8_t *)calloc(MAX_SB_SIZE * MAX_SB_SIZE, 1);
    mc->quarter_sb_buffer_stride   = MAX_SB_SIZE >> 1;
    mc->quarter_sb_buffer          = (uint8_t *)calloc((MAX_SB_SIZE >> 1) * (MAX_SB_SIZE >> 1), 1);
    mc->sixteenth_sb_buffer_stride = MAX_SB_SIZE >> 2;
    mc->sixteenth_sb_buffer        = (uint8_t *)calloc((MAX_SB_SIZE >> 2) * (MAX_SB_SIZE >> 2), 1);
    mc->interpolated_stride        = MAX_SEARCH_AREA_WIDTH;
    for (int l = 0; l < 2; l++) {
        mc->posb_buffer[l][0] = (uint8_t *)calloc((size_t)MAX_SEARCH_AREA_WIDTH * MAX_SEARCH_AREA_HEIGHT, 1);
        mc->posh_buffer[l][0] = (uint8_t *)calloc((size_t)MAX_SEARCH_AREA_WIDTH * MAX_SEARCH_AREA_HEIGHT, 1);
        mc->posj_buffer[l][0] = (uint8_t *)calloc((size_t)MAX_SEARCH_AREA_WIDTH * MAX_SEARCH_AREA_HEIGHT, 1);
    }
    mc->one_d_intermediate_results_buf0 = (uint8_t *)calloc(MAX_SB_SIZE * MAX_SB_SIZE, 1);
    mc->one_d_intermediate_results_buf1 = (uint8_t *)calloc(MAX_SB_SIZE * MAX_SB_SIZE, 1);
    mc->avctemp_buffer                  = (uint8_t *)calloc((size_t)MAX_SEARCH_AREA_WIDTH * MAX_SEARCH_AREA_HEIGHT, 1);
    mc->p_eight_pos_sad16x16            = (uint16_t *)calloc(8 * 16, sizeof(uint16_t));
    /* signals (Codec/EbMotionEstimationProcess.c:541-720) */
    mc->hme_search_type                     = HME_RECTANGULAR;
    mc->fractional_search_method            = p.fractional_search_method;
    mc->fractional_search_model             = p.fractional_search_model;
    mc->fractional_search64x64              = p.fractional_search64x64;
    mc->single_hme_quadrant                 = p.single_hme_quadrant;
    mc->search_area_width                   = p.search_area_width;
    mc->search_area_height                  = p.search_area_height;
    mc->number_hme_search_region_in_width   = p.number_hme_search_region_in_width;
    mc->number_hme_search_region_in_height  = p.number_hme_search_region_in_height;
    mc->hme_level0_total_search_area_width  = p.hme_level0_total_search_area_width;
    mc->hme_level0_total_search_area_height = p.hme_level0_total_search_area_height;
    for (int i = 0; i < 2; i++) {
        mc->hme_level0_search_area_in_width_array[i]  = p.hme_level0_search_area_in_width_array[i];
        mc->hme_level0_search_area_in_height_array[i] = p.hme_level0_search_area_in_height_array[i];
        mc->hme_level1_search_area_in_width_array[i]  = p.hme_level1_search_area_in_width_array[i];
        mc->hme_level1_search_area_in_height_array[i] = p.hme_level1_search_area_in_height_array[i];
        mc->hme_level2_search_area_in_width_array[i]  = p.hme_level2_search_area_in_width_array[i];
        mc->hme_level2_search_area_in_height_array[i] = p.hme_level2_search_area_in_height_array[i];
    }

    /* the SB loop of eb_vp9_motion_estimation_kernel (Codec/EbMotionEstimationProcess.c:964-1044) */
    EbPictureBufferDesc *in = pl[0], *q = pl[1], *s16 = pl[2];
    struct timespec t_begin, t_end;
    clock_gettime(CLOCK_MONOTONIC, &t_begin);
    for (int sb = sb_begin; sb < sb_end; sb++) {
        uint32_t ox = (uint32_t)(sb % nx) * 64, oy = (uint32_t)(sb / nx) * 64;
        uint32_t sw = (W - ox) < 64 ? W - ox : 64, sh = (H - oy) < 64 ? H - oy : 64;
        uint32_t bi = (in->origin_y + oy) * in->stride_y + in->origin_x + ox;
        for (uint32_t r = 0; r < 64; r++) memcpy(&mc->sb_buffer[r * 64], &in->buffer_y[bi + r * in->stride_y], 64);
        mc->sb_src_ptr    = &in->buffer_y[bi];
        mc->sb_src_stride = in->stride_y;
        if (pcs->enable_hme_level_1_flag) {
            uint32_t b = (q->origin_y + (oy >> 1)) * q->stride_y + q->origin_x + (ox >> 1);
            for (uint32_t r = 0; r < (sh >> 1); r++)
                memcpy(&mc->quarter_sb_buffer[r * mc->quarter_sb_buffer_stride], &q->buffer_y[b + r * q->stride_y], sw >> 1);
        }
        if (pcs->enable_hme_level_0_flag) {
            uint32_t b     = (s16->origin_y + (oy >> 2)) * s16->stride_y + s16->origin_x + (ox >> 2);
            uint8_t *fp    = &s16->buffer_y[b];
            uint8_t *local = mc->sixteenth_sb_buffer;
            for (uint32_t r = 0; r < (sh >> 2); r += 2) {
                memcpy(local, fp, sw >> 2);
                local += 16;
                fp += s16->stride_y << 1;
            }
        }
        motion_estimate_sb(pcs, (uint32_t)sb, ox, oy, mc, in);
    }
    clock_gettime(CLOCK_MONOTONIC, &t_end);
    const double loop_seconds = (double)(t_end.tv_sec - t_begin.tv_sec) + 1e-9 * (double)(t_end.tv_nsec - t_begin.tv_nsec);

    FILE *o = fopen(argv[2], "wb");
    if (!o) return 2;
    int32_t n = n_sb;
    fwrite(&n, 4, 1, o);
    for (int sb = 0; sb < n_sb; sb++)
        for (int pu = 0; pu < 85; pu++) {
            const MeCuResults *m = &pcs->me_results[sb][pu];
            svt_me_pu_result   r;
            memset(&r, 0, sizeof r);
            r.x_mv_l0 = m->x_mv_l0; r.y_mv_l0 = m->y_mv_l0; r.x_mv_l1 = m->x_mv_l1; r.y_mv_l1 = m->y_mv_l1;
            for (int c = 0; c < 3; c++) {
                r.distortion_direction[c].distortion = m->distortion_direction[c].distortion;
                r.distortion_direction[c].direction  = m->distortion_direction[c].direction;
            }
            r.total_me_candidate_index = m->total_me_candidate_index;
            fwrite(&r, sizeof r, 1, o);
        }
    fwrite(pcs->rcme_distortion, sizeof(uint32_t), (size_t)n_sb, o);
    fwrite(&loop_seconds, sizeof loop_seconds, 1, o);
    fclose(o);
    return 0;
}
