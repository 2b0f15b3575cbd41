/*
 * integration/coding_loop_binding.h -- the reference-side binding of libsvtvp9_hip.so's transform stage.
 *
 * What a maintainer of the reference adds to the encode pass (Source/Lib/Codec/EbEncDecProcess.c:3830, 3890, 3940: perform_coding_loop
 * for the Y, Cb and Cr transform block of every coded block; the function itself :365-587): instead of transforming a block where it
 * stands, svt_hip_bind_coding_loop APPENDS a descriptor (same arguments as perform_coding_loop minus the coefficient scratch), and
 * svt_hip_bind_coding_loop_flush runs the whole list -- an SB row, or a picture -- in one call.  Written against the reference's
 * EncDecContext / MACROBLOCKD / QUANTS; compiled (-Wall -Werror) and EXECUTED against them by oracle/ref_tqbind_driver.c, which calls the
 * reference's own perform_coding_loop for every block and compares coefficients, eobs and reconstruction (tests/test_ref_coding_loop.py).
 * INTEGRATION.md quotes this file.
 *
 * Include it after the reference's headers (EbEncDecProcess.h, vp9_blockd.h, vp9_scan.h, vp9_quantize.h) and after svtvp9_hip.h.
 */
#ifndef SVT_HIP_CODING_LOOP_BINDING_H
#define SVT_HIP_CODING_LOOP_BINDING_H
#include <stdlib.h>
#include <string.h>

/* per EncDec thread, beside its EncDecContext: the device context, the block list, and where the picture's planes and coefficients live.
 * The three planes of a set (source / prediction / reconstruction) are addressed as byte offsets from the set's base (the lowest of its
 * three plane pointers: the reference allocates a picture's planes in one EbPictureBufferDesc). */
typedef struct SvtHipTqBinding {
    svt_hip_ctx   *hip;
    svt_tq_block  *list;      /* capacity entries */
    int32_t        count, capacity;
    uint8_t       *src_base, *pred_base, *recon_base;
    size_t         plane_bytes; /* bytes from a base to the end of its last plane */
    int16_t       *qcoeff, *dqcoeff; /* coeff_capacity elements each: block i's n*n coefficients at list[i].coeff_off */
    uint16_t      *eob;       /* capacity entries, in list order */
    size_t         coeff_pos, coeff_capacity;
    svt_quant_tables qt[2];   /* luma, chroma of the picture's q index */
} SvtHipTqBinding;

/* the [DC, AC] pairs perform_coding_loop's callers pass (quants->y_zbin[q_index] ... &cpi->y_dequant[q_index][0], :3836-3843) */
static inline void svt_hip_bind_quant_tables(svt_quant_tables *t, const int16_t *zbin_ptr, const int16_t *round_ptr, const int16_t *quant_ptr,
                                             const int16_t *quant_shift_ptr, const int16_t *dequant_ptr) {
    for (int i = 0; i < 2; i++) {
        t->zbin[i] = zbin_ptr[i]; t->round[i] = round_ptr[i]; t->quant[i] = quant_ptr[i]; t->quant_shift[i] = quant_shift_ptr[i]; t->dequant[i] = dequant_ptr[i];
    }
}

/* where perform_coding_loop(context_ptr, ..., input_buffer, input_stride, pred_buffer, pred_stride, ..., recon_buffer, recon_stride, ...,
 * tx_size, plane, is_encode_pass = 1, do_recon) is called today.  Returns the block's index in the list (its eob is eob[index] after the
 * flush) or a negative value when the list is full. */
static inline int32_t svt_hip_bind_coding_loop(SvtHipTqBinding *b, EncDecContext *context_ptr, EbByte input_buffer, uint16_t input_stride, EbByte pred_buffer,
                                               uint16_t pred_stride, EbByte recon_buffer, uint16_t recon_stride, TX_SIZE tx_size, int plane, EB_BOOL do_recon) {
    if (b->count >= b->capacity || b->coeff_pos + ((size_t)16 << (2 * tx_size)) > b->coeff_capacity) return -1;
    /* transform type as perform_coding_loop derives it (:370-388) */
    MACROBLOCKD *const xd = context_ptr->e_mbd;
    TX_TYPE            tx_type = DCT_DCT;
    if (tx_size == TX_4X4) tx_type = get_tx_type_4x4(get_plane_type(plane), xd, context_ptr->bmi_index);
    else if (tx_size != TX_32X32) tx_type = get_tx_type(get_plane_type(plane), xd);
    const uint32_t *iscan_off = NULL;
    (void)svt_hip_vp9_iscan_tables(&iscan_off, NULL);
    svt_tq_block *k = &b->list[b->count];
    memset(k, 0, sizeof *k);
    k->src_off = (uint32_t)(input_buffer - b->src_base); k->pred_off = (uint32_t)(pred_buffer - b->pred_base); k->recon_off = (uint32_t)(recon_buffer - b->recon_base);
    k->src_stride = input_stride; k->pred_stride = pred_stride; k->recon_stride = recon_stride;
    k->coeff_off = (uint32_t)b->coeff_pos; b->coeff_pos += (size_t)16 << (2 * tx_size); /* n * n coefficients, contiguous */
    k->tx_size = (uint8_t)tx_size; k->tx_type = (uint8_t)tx_type;
    k->iscan_off = iscan_off[tx_size * 4 + (tx_size == TX_32X32 ? 0 : tx_type)];
    k->qtab = (uint8_t)(plane ? 1 : 0);
    k->do_recon = (uint8_t)(do_recon ? 1 : 0);
    return b->count++;
}

/* one call for everything appended: the library wants the blocks grouped by transform size, so the list is ordered (stable) and the
 * eobs are put back in append order.  Afterwards: qcoeff / dqcoeff at every block's coeff_off (what the reference writes to
 * residual_quant_coeff_buffer / recon_coeff_buffer), eob[i], and the reconstruction in the recon planes. */
static inline int svt_hip_bind_coding_loop_flush(SvtHipTqBinding *b) {
    const int32_t n = b->count;
    if (n == 0) return 0;
    svt_tq_block *sorted = (svt_tq_block *)malloc((size_t)n * sizeof *sorted);
    int32_t      *where  = (int32_t *)malloc((size_t)n * sizeof *where);
    uint16_t     *e      = (uint16_t *)malloc((size_t)n * sizeof *e);
    if (!sorted || !where || !e) { free(sorted); free(where); free(e); return -2; }
    int32_t at = 0;
    for (int ts = 0; ts < 4; ts++)
        for (int32_t i = 0; i < n; i++)
            if (b->list[i].tx_size == ts) { sorted[at] = b->list[i]; where[at++] = i; }
    const uint32_t *iscan_off = NULL;
    int32_t         iscan_entries = 0;
    const int16_t  *iscan = svt_hip_vp9_iscan_tables(&iscan_off, &iscan_entries);
    const int rc = svt_hip_tq_batch(b->hip, b->src_base, b->pred_base, b->recon_base, b->plane_bytes, sorted, n, b->qt, 2, iscan, (size_t)iscan_entries,
                                    b->qcoeff, b->dqcoeff, b->coeff_pos, e);
    if (rc == 0)
        for (int32_t j = 0; j < n; j++) b->eob[where[j]] = e[j];
    free(sorted); free(where); free(e);
    b->count = 0; b->coeff_pos = 0;
    return rc;
}
#endif
