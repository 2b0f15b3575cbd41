/*
 * ref_meside_driver.c -- harness that runs the REFERENCE's per-SB side decisions of the ME kernel
 * (Source/Lib/Codec/EbMotionEstimationProcess.c): compute_zz_sad (:431-534) and eb_vp9_derive_similar_collocated_flag
 * (:747-783).  TEST INFRASTRUCTURE ONLY (rules: ref_me_driver.c).  The harness allocates the control-set objects the two
 * functions read and fills exactly the fields they read.
 *
 * request : int32 magic 'SVMS', W, H, input_resolution, is_i_slice, is_used_as_reference,
 *           cur 1/16 plane {int32 stride, origin_x, origin_y, rows} + bytes, previous full plane {same} + bytes,
 *           int32 n_sb, cur_mean u8[n_sb], cur_var u16[n_sb], ref_mean u8[n_sb], ref_var u16[n_sb]
 * response: non_moving_index u8[n_sb], similar u8[n_sb], similar_all_layers u8[n_sb]
 *
 * Second request kind (stationary-edge flags; the two functions are static in the reference file, which this harness
 * therefore compiles as part of its own translation unit instead of linking its object):
 * request : int32 magic 'SVMT', W, H, input_resolution, temporal_layer_index, slice_type (0 B, 1 P, 2 I), run_part2,
 *           then per SB {int16 x_mv_l0, y_mv_l0; uint32 distortion; uint16 variance of the four 32x32 blocks}
 * response: per SB {check1, pm_check1, check2, low_dist_logo, potential_logo_sb, is_complete_sb} bytes; the SB geometry comes
 *           from the reference's eb_vp9_sb_params_init (Codec/EbSequenceControlSet.c:281)
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>

#include "EbDefinitions.h"
#include "EbPictureControlSet.h"
#include "EbSequenceControlSet.h"
#include "EbMotionEstimationProcess.h"
#include "EbMotionEstimationContext.h"
#include "EbReferenceObject.h"
#include "EbPictureBufferDesc.h"

/* the globals Codec/EbEncHandle.c defines in the encoder: the dispatch selector and the allocation-tracking table that
 * EB_MALLOC (used by eb_vp9_sb_params_init) writes into -- storage only, set up in main() */
uint32_t          eb_vp9_ASM_TYPES = 0;
EbMemoryMapEntry *memory_map       = 0;
uint32_t         *memory_map_index = 0;
uint64_t         *total_lib_memory = 0;
uint32_t          lib_malloc_count = 0;
static uint32_t   map_index_storage;
static uint64_t   total_memory_storage;

/* the reference's file itself: compute_zz_sad, eb_vp9_derive_similar_collocated_flag and the static
 * stationary_edge_over_update_over_time_sb_part1 / _part2 */
#include "EbMotionEstimationProcess.c"

static int rd(FILE *f, void *p, size_t n) { return fread(p, 1, n, f) == n ? 0 : -1; }
static int rd_plane(FILE *f, EbPictureBufferDesc *d) {
    int32_t g[4];
    if (rd(f, g, sizeof g)) return -1;
    d->stride_y = (uint16_t)g[0]; d->origin_x = (uint16_t)g[1]; d->origin_y = (uint16_t)g[2];
    const size_t n = (size_t)g[0] * g[3];
    d->buffer_y = (EbByte)malloc(n);
    return rd(f, d->buffer_y, n);
}

int main(int argc, char **argv) {
    if (argc < 3) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    int32_t h[7];
    if (rd(f, h, 6 * sizeof(int32_t))) return 3;
    memory_map = (EbMemoryMapEntry *)calloc(4096, sizeof *memory_map);
    memory_map_index = &map_index_storage; total_lib_memory = &total_memory_storage;
    if (h[0] == 0x544D5653) { /* 'SVMT' */
        if (rd(f, h + 6, sizeof(int32_t))) return 3;
        const int W = h[1], H = h[2], n_sb = ((W + 63) / 64) * ((H + 63) / 64);
        SequenceControlSet      *scs = (SequenceControlSet *)calloc(1, sizeof *scs);
        PictureParentControlSet *pcs = (PictureParentControlSet *)calloc(1, sizeof *pcs);
        scs->luma_width = (uint16_t)W; scs->luma_height = (uint16_t)H; scs->input_resolution = (uint8_t)h[3];
        if (eb_vp9_sb_params_init(scs) != EB_ErrorNone) return 4;
        pcs->temporal_layer_index = (uint8_t)h[4];
        pcs->slice_type = h[5] == 0 ? B_SLICE : h[5] == 1 ? P_SLICE : I_SLICE;
        pcs->sb_stat_array = (SbStat *)calloc((size_t)n_sb, sizeof(SbStat));
        pcs->me_results = (MeCuResults **)calloc((size_t)n_sb, sizeof(MeCuResults *));
        pcs->variance = (uint16_t **)calloc((size_t)n_sb, sizeof(uint16_t *));
        for (int i = 0; i < n_sb; i++) {
            struct { int16_t mx, my; uint32_t dist; uint16_t var[4]; } r;
            if (rd(f, &r, 16)) return 3;
            pcs->me_results[i] = (MeCuResults *)calloc(85, sizeof(MeCuResults));
            pcs->me_results[i][0].x_mv_l0 = r.mx; pcs->me_results[i][0].y_mv_l0 = r.my;
            pcs->me_results[i][0].distortion_direction[0].distortion = r.dist;
            pcs->variance[i] = (uint16_t *)calloc(85, 2);
            for (int k = 0; k < 4; k++) pcs->variance[i][ME_TIER_ZERO_PU_32x32_0 + k] = r.var[k];
        }
        fclose(f);
        FILE *o = fopen(argv[2], "wb");
        if (!o) return 2;
        for (int i = 0; i < n_sb; i++) {
            memset(&pcs->sb_stat_array[i], 0, sizeof(SbStat));
            stationary_edge_over_update_over_time_sb_part1(scs, pcs, (uint32_t)i);
            if (h[6]) stationary_edge_over_update_over_time_sb_part2(scs, pcs, (uint32_t)i);
            const SbStat *s = &pcs->sb_stat_array[i];
            uint8_t b[6] = {s->check1_for_logo_stationary_edge_over_time_flag, s->pm_check1_for_logo_stationary_edge_over_time_flag,
                            s->check2_for_logo_stationary_edge_over_time_flag, s->low_dist_logo, scs->sb_params_array[i].potential_logo_sb,
                            scs->sb_params_array[i].is_complete_sb};
            fwrite(b, 1, 6, o);
        }
        fclose(o);
        return 0;
    }
    if (h[0] != 0x534D5653) return 3; /* 'SVMS' */
    const int W = h[1], H = h[2];
    const int nx = (W + 63) / 64, ny = (H + 63) / 64, n_sb = nx * ny;

    SequenceControlSet      *scs  = (SequenceControlSet *)calloc(1, sizeof *scs);
    PictureParentControlSet *pcs  = (PictureParentControlSet *)calloc(1, sizeof *pcs), *prev = (PictureParentControlSet *)calloc(1, sizeof *prev);
    MotionEstimationContext *mec  = (MotionEstimationContext *)calloc(1, sizeof *mec);
    MeContext               *me   = (MeContext *)calloc(1, sizeof *me);
    EbObjectWrapper         *wprev = (EbObjectWrapper *)calloc(1, sizeof *wprev), *wref = (EbObjectWrapper *)calloc(1, sizeof *wref);
    EbPaReferenceObject     *ref  = (EbPaReferenceObject *)calloc(1, sizeof *ref);
    EbPictureBufferDesc      sixteenth, prev_input;
    memset(&sixteenth, 0, sizeof sixteenth); memset(&prev_input, 0, sizeof prev_input);
    if (rd_plane(f, &sixteenth) || rd_plane(f, &prev_input)) return 3;

    scs->picture_width_in_sb = (uint8_t)nx; scs->input_resolution = (uint8_t)h[3];
    scs->sb_params_array     = (SbParams *)calloc((size_t)n_sb, sizeof(SbParams));
    for (int y = 0; y < ny; y++)
        for (int x = 0; x < nx; x++) { /* as sb_params_init (Codec/EbSequenceControlSet.c) */
            SbParams *p = &scs->sb_params_array[y * nx + x];
            p->origin_x = (uint16_t)(x * 64); p->origin_y = (uint16_t)(y * 64);
            p->width  = (uint8_t)(W - x * 64 < 64 ? W - x * 64 : 64);
            p->height = (uint8_t)(H - y * 64 < 64 ? H - y * 64 : 64);
            p->is_complete_sb = (uint8_t)(p->width == 64 && p->height == 64);
        }
    me->sixteenth_sb_buffer = (uint8_t *)malloc(16 * 16); me->sixteenth_sb_buffer_stride = 16;
    mec->me_context_ptr = me;
    prev->enhanced_picture_ptr = &prev_input;
    prev->non_moving_index_array = (uint8_t *)calloc((size_t)n_sb, 1);
    wprev->object_ptr = prev;
    pcs->previous_picture_control_set_wrapper_ptr = wprev;
    if (compute_zz_sad(mec, scs, pcs, &sixteenth, 0, (uint32_t)nx, 0, (uint32_t)ny) != EB_ErrorNone) return 4;

    int32_t n;
    if (rd(f, &n, 4) || n != n_sb) return 3;
    uint8_t  *cm = (uint8_t *)malloc((size_t)n), *rm = (uint8_t *)malloc((size_t)n);
    uint16_t *cv = (uint16_t *)malloc(2 * (size_t)n), *rv = (uint16_t *)malloc(2 * (size_t)n);
    if (rd(f, cm, (size_t)n) || rd(f, cv, 2 * (size_t)n) || rd(f, rm, (size_t)n) || rd(f, rv, 2 * (size_t)n)) return 3;
    fclose(f);
    pcs->slice_type = h[4] ? I_SLICE : B_SLICE;
    pcs->is_used_as_reference_flag = (EB_BOOL)h[5];
    pcs->y_mean   = (uint8_t **)calloc((size_t)n, sizeof(uint8_t *));
    pcs->variance = (uint16_t **)calloc((size_t)n, sizeof(uint16_t *));
    pcs->similar_colocated_sb_array            = (EB_BOOL *)calloc((size_t)n, sizeof(EB_BOOL));
    pcs->similar_colocated_sb_array_all_layers = (EB_BOOL *)calloc((size_t)n, sizeof(EB_BOOL));
    for (int i = 0; i < n; i++) {
        pcs->y_mean[i] = (uint8_t *)calloc(85, 1); pcs->variance[i] = (uint16_t *)calloc(85, 2);
        pcs->y_mean[i][PA_RASTER_SCAN_CU_INDEX_64x64] = cm[i]; pcs->variance[i][PA_RASTER_SCAN_CU_INDEX_64x64] = cv[i];
        ref->y_mean[i] = rm[i]; ref->variance[i] = rv[i];
    }
    wref->object_ptr = ref;
    pcs->ref_pa_pic_ptr_array[REF_LIST_0] = wref;
    for (int i = 0; i < n; i++) eb_vp9_derive_similar_collocated_flag(pcs, (uint32_t)i);

    FILE *o = fopen(argv[2], "wb");
    if (!o) return 2;
    fwrite(prev->non_moving_index_array, 1, (size_t)n, o);
    for (int i = 0; i < n; i++) { uint8_t b = (uint8_t)pcs->similar_colocated_sb_array[i]; fwrite(&b, 1, 1, o); }
    for (int i = 0; i < n; i++) { uint8_t b = (uint8_t)pcs->similar_colocated_sb_array_all_layers[i]; fwrite(&b, 1, 1, o); }
    fclose(o);
    return 0;
}
