/*
 * gop_shard.c -- host-side (plain C) work distribution of the hot path over the GPUs of a node (scope row e).
 * Closed GOPs share nothing (Codec/EbPictureDecisionProcess.c:952, 1596-1603): round-robin by GOP index; a split GOP hands one
 * padded reference picture from a mini-GOP's device to the next one's (svt_hip_ref_handoff_device).
 */
#include "../../include/svtvp9_hip.h"

int32_t svt_hip_gop_owner(int64_t gop, int32_t n_devices) { return n_devices > 0 && gop >= 0 ? (int32_t)(gop % n_devices) : -1; }

int32_t svt_hip_gop_assign(int64_t n_gops, int32_t n_devices, int32_t index, int64_t *gops, int32_t max) {
    if (n_gops < 0 || n_devices < 1 || index < 0 || index >= n_devices || (max > 0 && !gops)) return SVT_HIP_ERR_BAD_PARAMETER;
    int32_t n = 0;
    for (int64_t g = index; g < n_gops; g += n_devices) {
        if (n < max) gops[n] = g;
        n++;
    }
    return n;
}

int32_t svt_hip_minigop_reference_source(int64_t minigop, int32_t n_devices) {
    if (n_devices < 1 || minigop <= 0) return -1;
    return (int32_t)((minigop - 1) % n_devices);
}

/* The mini-GOP table of Codec/EbUtility.c:167-185 (hierarchical levels, first and last picture, length) restricted to the 16-picture
 * window the picture-decision kernel looks at, and the walk of eb_vp9_generate_picture_window_split (:387-430) over it with the
 * activity flags the kernel sets (:1665-1675): the 16-picture entry is a candidate only for a full buffer, the two 8-picture
 * entries otherwise; what the walk leaves uncovered becomes one trailing part (eb_vp9_handle_incomplete_picture_window_map). */
int32_t svt_hip_minigop_split(int32_t n, int32_t levels, int32_t cut_by_intra, svt_minigop_part parts[4]) {
    static const struct { int lv, start, end, len; } tab[15] = {{5, 0, 31, 32}, {4, 0, 15, 16}, {3, 0, 7, 8}, {2, 0, 3, 4}, {2, 4, 7, 4}, {3, 8, 15, 8},
                                                                {2, 8, 11, 4}, {2, 12, 15, 4}, {4, 16, 31, 16}, {3, 16, 23, 8}, {2, 16, 19, 4},
                                                                {2, 20, 23, 4}, {3, 24, 31, 8}, {2, 24, 27, 4}, {2, 28, 31, 4}};
    static const int offset[4] = {1, 3, 7, 31};
    if (!parts || n < 1 || levels < 0 || levels > 5 || n > (1 << levels)) return -1;
    int np = 0;
    if (n == 1) { /* a single picture is never split (:1663) */
        parts[0].start = 0; parts[0].length = 1; parts[0].hierarchical_levels = levels; np = 1;
    } else {
        int active[15];
        for (int i = 0; i < 15; i++) active[i] = tab[i].lv != 2;
        if (n == 16) active[1] = 0; else { active[2] = 0; active[5] = 0; }
        for (int i = 0; i < 15;) {
            if (tab[i].end < n && !active[i] && np < 4) {
                parts[np].start = tab[i].start; parts[np].length = tab[i].len; parts[np].hierarchical_levels = tab[i].lv; np++;
            }
            i += active[i] ? 1 : offset[tab[i].lv - 2];
        }
        if (np == 0) { parts[0].start = 0; parts[0].length = n; parts[0].hierarchical_levels = 3; np = 1; }
        else if (parts[np - 1].start + parts[np - 1].length < n && np < 4) {
            parts[np].start = parts[np - 1].start + parts[np - 1].length; parts[np].length = n - parts[np].start; parts[np].hierarchical_levels = 3; np++;
        }
    }
    /* a part keeps the random-access hierarchy when it is a whole period of its own levels (:1711-1714: mini_gop_length <
       pred_struct_period switches to low-delay P); mini_gop_idr_count is non-zero for the LAST part only (:419-424, 463-472), the one
       that ends with the intra picture, and forces low-delay P there whatever its length */
    for (int i = 0; i < np; i++) parts[i].random_access = parts[i].length == (1 << parts[i].hierarchical_levels) && !(cut_by_intra && i == np - 1);
    return np;
}
