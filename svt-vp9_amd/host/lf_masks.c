/*
 * lf_masks.c -- host-side (plain C) construction of the per-SB LOOP_FILTER_MASKs from the mode-info grid.
 *
 * Mirrors eb_vp9_build_mask_frame (VPX/vp9_loopfilter.c:1548-1571).  The reference has two formulations of the same
 * rules: the frame-level tree walk eb_vp9_setup_mask (:901-1040) and the per-block eb_vp9_build_mask (:1587-1689)
 * used when masks are built in line.  This file follows the per-block one: every 8x8 unit that is the top-left unit
 * of its prediction block contributes the block's edges at shift = row*8 + col of its SB.
 *
 * The reference's constant tables (prediction / size / transform masks, uv transform size) are pure geometry; they are
 * computed from the block dimensions instead of being tabulated.  The per-SB rule itself lives in csrc/encdec_core.h
 * (svt_lf_mask_build_sb): the same text is compiled into the device-side builder the EncDec driver uses (svt_lf_mask_kernel), so what
 * the CPU tests pin here against the reference's eb_vp9_setup_mask is also what runs on the GPU.
 */
#include "../../include/svtvp9_hip.h"
#include "../csrc/encdec_core.h"

int32_t svt_hip_lf_build_masks(const svt_lf_mode_info *mi, int32_t mi_stride, int32_t mi_rows, int32_t mi_cols,
                               svt_lf_mask *lfm, int32_t lfm_stride) {
    if (!mi || !lfm || mi_rows < 1 || mi_cols < 1 || mi_stride < mi_cols || lfm_stride < (mi_cols + 7) / 8) return SVT_HIP_ERR_BAD_PARAMETER;
    for (int sb_r = 0; sb_r < (mi_rows + 7) / 8; sb_r++)
        for (int sb_c = 0; sb_c < (mi_cols + 7) / 8; sb_c++)
            if (svt_lf_mask_build_sb(mi, mi_stride, mi_rows, mi_cols, sb_r, sb_c, &lfm[sb_r * lfm_stride + sb_c])) return SVT_HIP_ERR_BAD_PARAMETER;
    return SVT_HIP_OK;
}
