/*
 * me_emu.c -- serial host emulation of the HIP ME kernel (svt-vp9_amd/csrc/me_core.h compiled with
 * SVT_HOST_EMU).  TEST INFRASTRUCTURE ONLY: it lets the CPU test-suite exercise the kernel's phase logic
 * against the oracle without a GPU.  It is never loaded by the product or by bench.py.
 */
#define SVT_HOST_EMU 1
#include <stdlib.h>
#include <string.h>
#include "../../svt-vp9_amd/csrc/me_core.h"
#include "../../svt-vp9_amd/csrc/me_layout.h"

static uint8_t *g_lds = 0;
void svt_emu_lds_fill(int off, int len, int val) { if (g_lds) memset(g_lds + off, val, (size_t)len); }
int  svt_emu_layout(const svt_me_params *p, me_lds_layout *L) { return me_lds_layout_compute(p, L); }
int32_t svt_emu_me_picture(const svt_pa_picture *cur, const svt_pa_picture *ref0, const svt_pa_picture *ref1,
                           const svt_me_params *params, svt_me_pu_result *results, uint32_t *rcme, int32_t sb_begin,
                           int32_t sb_end) {
    me_lds_layout L;
    if (me_lds_layout_compute(params, &L)) return -4;
    if (me_tables_selfcheck()) return -100 - me_tables_selfcheck();
    /* persistent "LDS": with SVT_EMU_POISON=keep the content left by the previous SB / call stays, exactly
       like real LDS inherited from the previous workgroup on that CU */
    uint8_t *lds = g_lds;
    if (!lds) { lds = (uint8_t *)aligned_alloc(16, 160 * 1024 + 64); memset(lds, 0, 160 * 1024); g_lds = lds; }
    me_pic_dev pic;
    memset(&pic, 0, sizeof pic);
    pic.cur = *cur; pic.ref[0] = *ref0; if (ref1) pic.ref[1] = *ref1;
    pic.results = results; pic.rcme = rcme;
    int W = cur->full.width, H = cur->full.height, nx = (W + 63) / 64, ny = (H + 63) / 64;
    if (sb_end < 0 || sb_end > nx * ny) sb_end = nx * ny;
    for (int sb = sb_begin; sb < sb_end; sb++) {
        /* poison: nothing may depend on stale LDS.  SVT_EMU_POISON selects the pattern: a byte value, or
           "src" = bytes of the source picture (realistic stale data that could win a search) */
        {
            const char *pz = getenv("SVT_EMU_POISON");
            if (pz && !strcmp(pz, "keep")) {
            } else if (pz && !strcmp(pz, "src")) {
                const uint8_t *sp = cur->full.buf + (size_t)cur->full.origin_y * cur->full.stride;
                size_t         n  = (size_t)cur->full.stride * cur->full.height;
                for (int i = 0; i < L.total_bytes; i++) lds[i] = sp[((size_t)i * 7 + (size_t)sb * 131) % n];
            } else {
                memset(lds, pz ? atoi(pz) : 0xA5, (size_t)L.total_bytes);
            }
        }
        me_ctx_t c;
        c.pic = &pic; c.p = params; c.L = L; c.lds = lds;
        c.st = (me_state_t *)(lds + L.off_state); c.src = lds + L.off_src; c.region = lds + L.off_region;
        c.planes = lds + L.off_planes; c.hme_scratch = lds + L.off_region; c.hme_scratch_bytes = (L.off_planes - L.off_region) + L.scratch_bytes; c.quarter_sb = lds + L.off_quarter; c.ssdc = params->fractional_search_method == SVT_SSD_SEARCH ? (uint32_t *)(lds + L.off_ssd) : 0; c.pred0 = (uint32_t *)(lds + L.off_pred0); c.cand = (uint32_t *)(lds + L.off_cand); c.cand_hi = (uint32_t *)(lds + (L.off_cand_hi >= 0 ? L.off_cand_hi : L.off_cand)); c.redo = 0;
        c.pic_w = W; c.pic_h = H; c.sb_index = sb; c.prof = 0;
        c.sb_x = (sb % nx) * 64; c.sb_y = (sb / nx) * 64;
        c.sb_w = (W - c.sb_x) < 64 ? W - c.sb_x : 64; c.sb_h = (H - c.sb_y) < 64 ? H - c.sb_y : 64;
        me_sb_run(&c, 0);
    }
    return 0;
}
