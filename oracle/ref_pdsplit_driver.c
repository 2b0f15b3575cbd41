/*
 * ref_pdsplit_driver.c -- harness that runs the REFERENCE's mini-GOP window split: eb_vp9_initialize_mini_gop_activity_array,
 * eb_vp9_generate_picture_window_split and eb_vp9_handle_incomplete_picture_window_map (Source/Lib/Codec/EbPictureDecisionProcess.c:
 * 367-476) over the reference's own mini-GOP table (Codec/EbUtility.c:167-192), for every pre-assignment buffer size 2..16.
 * TEST INFRASTRUCTURE ONLY (rules: ref_me_driver.c).  The three functions are called in the order and with the two activity
 * flags the picture-decision kernel sets between them (:1662-1680; those assignments are inline in the thread function and are
 * repeated here as the call sequence, they are not arithmetic).
 *
 * output (stdout): one line per buffer size n: "n parts  start length levels  ..."
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "EbDefinitions.h"
#include "EbUtility.h"
#include "EbPictureControlSet.h"
#include "EbSequenceControlSet.h"
#include "EbPictureDecisionProcess.h"
#include "EbEncodeContext.h"

int main(void) {
    PictureDecisionContext *ctx = (PictureDecisionContext *)calloc(1, sizeof *ctx);
    EncodeContext          *enc = (EncodeContext *)calloc(1, sizeof *enc);
    for (uint32_t n = 2; n <= 16; n++) {
        enc->pre_assignment_buffer_count = n;
        enc->pre_assignment_buffer_intra_count = 0;
        enc->pre_assignment_buffer_idr_count = 0;
        ctx->total_number_of_mini_gops = 1;
        eb_vp9_initialize_mini_gop_activity_array(ctx);
        if (n == 16) ctx->mini_gop_activity_array[L5_0_INDEX] = EB_FALSE;
        else { ctx->mini_gop_activity_array[L4_0_INDEX] = EB_FALSE; ctx->mini_gop_activity_array[L4_1_INDEX] = EB_FALSE; }
        eb_vp9_generate_picture_window_split(ctx, enc);
        eb_vp9_handle_incomplete_picture_window_map(ctx, enc);
        printf("%u %u", n, ctx->total_number_of_mini_gops);
        for (uint32_t i = 0; i < ctx->total_number_of_mini_gops; i++)
            printf("  %u %u %u", ctx->mini_gop_start_index[i], ctx->mini_gop_length[i], ctx->mini_gop_hierarchical_levels[i]);
        printf("\n");
    }
    return 0;
}
