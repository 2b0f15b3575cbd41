/* me_layout.h -- host-side computation of the ME kernel's LDS layout from the search parameters. */
#ifndef SVT_ME_LAYOUT_H
#define SVT_ME_LAYOUT_H
#include "me_core.h"

#if defined(__HIPCC__)
#define ME_LAYOUT_FN __host__ __device__ static inline
#else
#define ME_LAYOUT_FN static inline
#endif
ME_LAYOUT_FN int me_round_up(int v, int a) { return (v + a - 1) / a * a; }

/* The geometry: sizes follow the largest search area the parameters allow (clipping at picture borders only shrinks it).
 * Also callable in the kernel: a specialised instance evaluates it on its compile-time parameters, so strides and offsets
 * are immediates instead of scalar registers (the kernel is short of those). */
ME_LAYOUT_FN void me_lds_layout_geom(const svt_me_params *p, me_lds_layout *L) {
    int saw = p->search_area_width < 127 ? p->search_area_width : 127;
    int sah = p->search_area_height < 127 ? p->search_area_height : 127;
    if (saw < 1) saw = 1;
    if (sah < 1) sah = 1;
    int W = saw + ME_SB - 1, H = sah + ME_SB - 1;
    int rs = me_round_up(W + ME_RGN_GX + 4 + 16, 4); /* (+ 16: the last 16-byte unit of a row of a clipped area) */
    if (((rs >> 2) & 1) == 0) rs += 4; /* odd number of dwords per row: rows spread over LDS banks */
    L->region_stride = rs;
    L->region_rows   = H + 2 * ME_RGN_GY + 1;
    int pd = (W + 2 * ME_PL_G + 3) >> 2; /* plane dwords per row (ph_interp_bh); kept odd like the region's */
    pd |= 1;
    L->plane_stride  = 4 * pd;
    /* + 16: the aligned-dword fetches of me_block_sad_rows may touch the dword after a row's last sample */
    L->plane_bytes   = me_round_up((H + 2 * ME_PL_G) * L->plane_stride + 16, 16);
    int off          = 0;
    L->off_state     = off; off += me_round_up((int)sizeof(me_state_t), 16);
    L->off_src       = off; off += ME_SB * ME_SB;
    L->off_region    = off; off += me_round_up(L->region_rows * rs, 16);
    L->off_planes    = off; L->scratch_bytes = 3 * L->plane_bytes; off += L->scratch_bytes;
    L->off_quarter   = off; if (p->enable_hme_level_1_flag) off += 32 * 32;
    L->off_ssd       = off; if (p->fractional_search_method == SVT_SSD_SEARCH) off += me_round_up(85 * 10 * 4, 16);
    /* cu8x8_mode == 1: PUs 21..84 are neither refined (me_pu_refined) nor bi-predicted (me_pu_bipred) */
    L->cand_dwords   = 8 * (p->cu8x8_mode == 1 ? 21 : 85);
    /* a short table lives in the bytes of the state's first union (full-pel keys / HME work list: neither is live while the
     * sub-pel and bi-pred candidates are) */
    if ((size_t)L->cand_dwords * 4 <= sizeof(((me_state_t *)0)->key)) L->off_cand = L->off_state;
    else { L->off_cand = off; off += me_round_up(L->cand_dwords * 4, 16); }
    L->off_pred0     = off;
#ifdef SVT_HOST_EMU /* the kernel keeps list 0's prediction dwords in registers */
    if (p->num_ref_lists == 2) off += 16 * 256 * 4;
#endif
    L->total_bytes = off;
}

/* Geometry + the HME level-0 area multipliers.  Returns 0, or -1 when the configuration does not fit in 160 KiB of LDS. */
static inline int me_lds_layout_compute(const svt_me_params *p, me_lds_layout *L) {
    me_lds_layout_geom(p, L);
    {   /* HME level-0 search area multipliers, Codec/EbDefinitions.h:989-1005, indexed [hierarchical_levels][temporal_layer] */
        static const int32_t mult_tab[6][6] = {{100, 0, 0, 0, 0, 0},       {100, 100, 0, 0, 0, 0},
                                               {100, 100, 100, 0, 0, 0},   {200, 140, 100, 70, 0, 0},
                                               {350, 200, 100, 100, 100, 0}, {525, 350, 200, 100, 100, 100}};
        const int hl = p->hierarchical_levels < 6 ? p->hierarchical_levels : 5, tl = p->temporal_layer_index < 6 ? p->temporal_layer_index : 5;
        const int mult = mult_tab[hl][tl];
        for (int i = 0; i < 2; i++) {
            L->hme_w0[i] = (int16_t)((p->hme_level0_search_area_in_width_array[i] * mult) / 100);
            L->hme_h0[i] = (int16_t)((p->hme_level0_search_area_in_height_array[i] * mult) / 100);
        }
        L->hme_tw0 = (int16_t)((p->hme_level0_total_search_area_width * mult) / 100);
        L->hme_th0 = (int16_t)((p->hme_level0_total_search_area_height * mult) / 100);
    }
    return L->total_bytes <= 160 * 1024 ? 0 : -1;
}
#endif
