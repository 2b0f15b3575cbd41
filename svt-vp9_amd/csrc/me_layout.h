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
/* compact = 1 (search areas whose width is a multiple of 8 only): no room for the 16 tail columns a clipped area's row may need -- a
 * workgroup that meets such an area leaves it to a second launch with the full layout (me_ctx_t::redo) -- and the quarter-resolution SB
 * shares the bytes of the SSD tables (dead during HME; reloaded for list 1).  Buys the 64 x 64-area configurations a second workgroup
 * per CU: 82 336 / 83 360 -> 79 168 / 80 192 bytes (cu8x8_mode 1 / 0) against the 81 920 two workgroups may take each. */
ME_LAYOUT_FN void me_lds_layout_geom_ex(const svt_me_params *p, me_lds_layout *L, int compact) {
    int saw = p->search_area_width < 127 ? p->search_area_width : 127;
    int sah = p->search_area_height < 127 ? p->search_area_height : 127;
    if (saw < 1) saw = 1;
    if (sah < 1) sah = 1;
    int W = saw + ME_SB - 1, H = sah + ME_SB - 1;
    int rs = me_round_up(W + ME_RGN_GX + 4 + (compact ? 0 : 16), 4); /* (+ 16: the last 16-byte unit of a row of a clipped area) */
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
    L->off_ssd       = off; if (p->fractional_search_method == SVT_SSD_SEARCH) off += me_round_up(85 * 9 * 4, 16);
    if (compact && p->enable_hme_level_1_flag && p->fractional_search_method == SVT_SSD_SEARCH) { /* the 1024 bytes of the quarter SB inside the SSD tables */
        off -= 32 * 32; L->off_ssd = L->off_quarter;
    }
    L->compact       = compact;
    /* cu8x8_mode == 1: PUs 21..84 are neither refined (me_pu_refined) nor bi-predicted (me_pu_bipred) */
    L->cand_dwords   = 8 * (p->cu8x8_mode == 1 ? 21 : 85);
    /* the dword entries (PUs 0..20; bi-prediction sums [pu]) live in the bytes of the state's first union (full-pel keys / HME work
     * list: neither is live while the sub-pel and bi-pred candidates are); the halfword entries of the 8x8 PUs behind the tables */
    L->off_cand      = L->off_state; /* 168 dwords <= sizeof key[85] */
    L->off_cand_hi   = -1;
    if (L->cand_dwords > 168) { L->off_cand_hi = off; off += (L->cand_dwords - 168) * 2; }
    L->off_pred0     = off;
#ifdef SVT_HOST_EMU /* the kernel keeps list 0's prediction dwords in registers */
    if (p->num_ref_lists == 2) off += 16 * 256 * 4;
#endif
    L->total_bytes = off;
}

ME_LAYOUT_FN void me_lds_layout_geom(const svt_me_params *p, me_lds_layout *L) { me_lds_layout_geom_ex(p, L, 0); }

/* Geometry + the HME level-0 area multipliers.  Returns 0, or -1 when the configuration does not fit in 160 KiB of LDS. */
static inline int me_lds_layout_compute_ex(const svt_me_params *p, me_lds_layout *L, int compact) {
    me_lds_layout_geom_ex(p, L, compact);
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
static inline int me_lds_layout_compute(const svt_me_params *p, me_lds_layout *L) { return me_lds_layout_compute_ex(p, L, 0); }
/* workgroups per CU of a layout (LDS alone: 160 KiB in granules of 1280 bytes) */
static inline int me_lds_workgroups_per_cu(const me_lds_layout *L) { return 128 / ((L->total_bytes + 1279) / 1280); }
#endif
