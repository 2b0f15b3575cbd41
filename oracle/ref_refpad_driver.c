/*
 * ref_refpad_driver.c -- harness that runs the REFERENCE's pad_ref_and_set_flags (Source/Lib/Codec/EbEncDecProcess.c:4822-4851,
 * a static function: the harness compiles that file inside its own translation unit) -> eb_vp9_generate_padding
 * (Codec/EbMcp.c:17-58) on a reconstructed picture.  TEST INFRASTRUCTURE ONLY (rules: ref_me_driver.c).  The harness allocates
 * the control-set objects the function reads (picture control set -> parent -> reference_picture_wrapper_ptr -> EbReferenceObject
 * -> reference_picture) and fills exactly the fields it reads; nothing else of the encoder is entered.
 *
 * request 'SVFL' (round 4): the stage flags of the EncDec kernel -- eb_vp9_signal_derivation_enc_dec_kernel_{sq,oq,vmaf}
 *           (Codec/EbEncDecProcess.c:4912-5300) for every (tune 0..2, enc_mode 0..12, temporal layer 0..4, is_used_as_reference 0/1) on
 *           control sets holding exactly the fields they read; response: per tuple the two bytes limit_intra, allow_enc_dec_mismatch
 *           (pins svt_hip_encdec_flags_derive).
 * request : int32 magic 'SVRP', width, height, origin_x, origin_y, stride_y, stride_c, then the three padded buffers
 *           (Y: stride_y * (height + 2 origin_y) bytes, Cb and Cr: stride_c * (height / 2 + origin_y) bytes each) with
 *           arbitrary border content
 * response: the three buffers after padding
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>

#include "EbDefinitions.h"

uint32_t          eb_vp9_ASM_TYPES = 0; /* `-asm 0` */
EbMemoryMapEntry *memory_map       = 0;
uint32_t         *memory_map_index = 0;
uint64_t         *total_lib_memory = 0;
uint32_t          lib_malloc_count = 0;

#include "EbEncDecProcess.c"

static int rd(FILE *f, void *p, size_t n) { return fread(p, 1, n, f) == n ? 0 : -1; }

int main(int argc, char **argv) {
    if (argc < 3) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    int32_t h[7];
    if (rd(f, h, sizeof h[0])) return 3;
    if (h[0] == 0x4C465653) { /* 'SVFL' */
        fclose(f);
        FILE *o = fopen(argv[2], "wb");
        if (!o) return 2;
        for (int tune = 0; tune < 3; tune++)
            for (int mode = 0; mode <= 12; mode++)
                for (int layer = 0; layer < 5; layer++)
                    for (int used = 0; used < 2; used++) {
                        PictureParentControlSet *ppc = (PictureParentControlSet *)calloc(1, sizeof *ppc);
                        PictureControlSet       *pcs = (PictureControlSet *)calloc(1, sizeof *pcs);
                        SequenceControlSet      *scs = (SequenceControlSet *)calloc(1, sizeof *scs);
                        EncDecContext           *ctx = (EncDecContext *)calloc(1, sizeof *ctx);
                        pcs->parent_pcs_ptr = ppc;
                        pcs->enc_mode = ppc->enc_mode = (uint8_t)mode;
                        pcs->temporal_layer_index = ppc->temporal_layer_index = (uint8_t)layer;
                        ppc->is_used_as_reference_flag = (EB_BOOL)used;
                        pcs->slice_type = B_SLICE;
                        scs->input_resolution = INPUT_SIZE_4K_RANGE;
                        scs->static_config.tune = (uint8_t)tune;
                        if (tune == 0) eb_vp9_signal_derivation_enc_dec_kernel_sq(scs, pcs, ctx);
                        else if (tune == 2) eb_vp9_signal_derivation_enc_dec_kernel_vmaf(scs, pcs, ctx);
                        else eb_vp9_signal_derivation_enc_dec_kernel_oq(scs, pcs, ctx);
                        const uint8_t out[2] = {(uint8_t)ctx->limit_intra, (uint8_t)ctx->allow_enc_dec_mismatch};
                        fwrite(out, 1, 2, o);
                        free(ppc); free(pcs); free(scs); free(ctx);
                    }
        fclose(o);
        return 0;
    }
    if (rd(f, h + 1, 6 * sizeof h[0]) || h[0] != 0x50525653) return 3; /* 'SVRP' */
    EbPictureBufferDesc *pic = (EbPictureBufferDesc *)calloc(1, sizeof *pic);
    pic->width = (uint16_t)h[1]; pic->height = (uint16_t)h[2]; pic->origin_x = (uint16_t)h[3]; pic->origin_y = (uint16_t)h[4];
    pic->stride_y = (uint16_t)h[5]; pic->stride_cb = pic->stride_cr = (uint16_t)h[6];
    const size_t ny = (size_t)h[5] * (h[2] + 2 * h[4]), nc = (size_t)h[6] * (h[2] / 2 + h[4]);
    pic->buffer_y = (EbByte)malloc(ny); pic->buffer_cb = (EbByte)malloc(nc); pic->buffer_cr = (EbByte)malloc(nc);
    if (rd(f, pic->buffer_y, ny) || rd(f, pic->buffer_cb, nc) || rd(f, pic->buffer_cr, nc)) return 3;
    fclose(f);
    EbReferenceObject       *ro  = (EbReferenceObject *)calloc(1, sizeof *ro);
    EbObjectWrapper         *wr  = (EbObjectWrapper *)calloc(1, sizeof *wr);
    PictureParentControlSet *ppc = (PictureParentControlSet *)calloc(1, sizeof *ppc);
    PictureControlSet       *pcs = (PictureControlSet *)calloc(1, sizeof *pcs);
    SequenceControlSet      *scs = (SequenceControlSet *)calloc(1, sizeof *scs);
    ro->reference_picture = pic;
    wr->object_ptr = ro;
    ppc->reference_picture_wrapper_ptr = wr;
    ppc->is_used_as_reference_flag = EB_TRUE;
    pcs->parent_pcs_ptr = ppc;
    scs->static_config.encoder_bit_depth = EB_8BIT;
    pad_ref_and_set_flags(pcs, scs);
    FILE *o = fopen(argv[2], "wb");
    if (!o) return 2;
    fwrite(pic->buffer_y, 1, ny, o); fwrite(pic->buffer_cb, 1, nc, o); fwrite(pic->buffer_cr, 1, nc, o);
    fclose(o);
    return 0;
}
