/*
 * me_kernel.hip -- gfx950 kernels for the motion-estimation path and their launchers.
 * The per-SB algorithm lives in me_core.h (shared verbatim with the CPU test emulation).
 */
#include <hip/hip_runtime.h>
#include "me_core.h"
#include "me_fast.h"
#include "me_layout.h"
#include "me_spec.h"
#include "svt_ctx.h"
#include <stdio.h>
#include <stdlib.h>

extern __shared__ __align__(16) uint8_t svt_lds[];

#ifndef ME_WAVES_PER_EU
#define ME_WAVES_PER_EU 3 /* generic instance: 3 workgroups of 4 waves per CU: <= 168 VGPRs, <= 53 KB of LDS */
#endif
#ifndef ME_WAVES_PER_EU_SPEC
/* specialised instances are held to 96 VGPRs (no spills): the 2160p instance needs 31.9 KB of LDS = 25 allocation
 * granules of 1280 bytes, so five workgroups fit a CU = 5 waves per SIMD = 480 of its 512 registers.  (The 1080p / 360p
 * instances need ~36 KB: four workgroups, and then the 128 registers left on a SIMD are exactly one wave of the 32x32
 * transform kernel, of the deblocking kernel (96) or of the 16x16 transform kernel (80) beside them when the stages
 * overlap -- at 112 VGPRs per ME wave none of them fitted.) */
#define ME_WAVES_PER_EU_SPEC 5
#endif
/* One workgroup per (picture, SB).  blockIdx -> work item mapping is XCD-aware: consecutive work items
 * (neighbouring SBs, which share most of their reference window) are placed on the same XCD so that the
 * window re-reads hit that XCD's L2 (block b runs on XCD b % 8). */
/* COMPACT / redo (me_layout.h): the launch with the compact layout flags the SBs it cannot serve in redo[picture * n_sb + sb]; the launch with
 * the full layout that follows it (COMPACT = false, redo != nullptr) runs the flagged SBs only. */
template <int SPEC, bool FAST, bool COMPACT = false>
__device__ __forceinline__ void me_kernel_body(const me_pic_dev *__restrict__ pics, svt_me_params p, me_lds_layout L, int n_sb, int nx, int pic_w, int pic_h,
                                               int total, int chunk, unsigned long long *prof, uint32_t *redo = nullptr) {
    /* block b runs on XCD b & 7: XCD k takes the k-th eighth of the SBs of EVERY picture (chunk SBs each; pictures of different
     * temporal layers cost differently, an XCD per picture range would leave the XCDs unbalanced), in picture order */
    const int b = blockIdx.x, j = b >> 3;
    const int pic = j / chunk, sb = (b & 7) * chunk + (j - pic * chunk);
    if (sb >= n_sb || pic * n_sb >= total) return;
    if (!COMPACT && redo && !redo[pic * n_sb + sb]) return;
    me_ctx_t  c;
    c.redo = COMPACT ? redo + pic * n_sb + sb : nullptr;
    c.pic = &pics[pic];
    svt_me_params pp = p;
    /* per-picture parameters travel with the picture (a launch may mix temporal layers) */
    pp.num_ref_lists = c.pic->num_ref_lists; pp.temporal_layer_index = c.pic->temporal_layer_index;
    pp.hierarchical_levels = c.pic->hierarchical_levels; pp.same_ref_poc = c.pic->same_ref_poc;
    L.hme_w0[0] = c.pic->hme_w0[0]; L.hme_w0[1] = c.pic->hme_w0[1]; L.hme_h0[0] = c.pic->hme_h0[0]; L.hme_h0[1] = c.pic->hme_h0[1];
    L.hme_tw0 = c.pic->hme_tw0; L.hme_th0 = c.pic->hme_th0;
    if constexpr (SPEC != 0) me_spec_apply<SPEC>(&pp); /* constants equal to the caller's values (me_spec_match) */
    c.p   = &pp;
    if constexpr (SPEC != 0) { /* the same values as the host's, as compile-time constants */
        me_lds_layout G = L;
        me_lds_layout_geom_ex(&pp, &G, COMPACT);
        L.region_stride = G.region_stride; L.plane_stride = G.plane_stride; L.plane_bytes = G.plane_bytes; L.region_rows = G.region_rows;
        L.cand_dwords = G.cand_dwords; L.scratch_bytes = G.scratch_bytes;
        L.off_state = G.off_state; L.off_src = G.off_src; L.off_region = G.off_region; L.off_planes = G.off_planes;
        L.off_quarter = G.off_quarter; L.off_ssd = G.off_ssd; L.off_cand = G.off_cand; L.off_cand_hi = G.off_cand_hi; L.off_pred0 = G.off_pred0;
    }
    L.compact = COMPACT;
    c.L   = L;
    c.lds = svt_lds;
    c.st     = (me_state_t *)(svt_lds + L.off_state);
    c.src    = svt_lds + L.off_src;
    c.region = svt_lds + L.off_region;
    c.planes = svt_lds + L.off_planes;
    c.hme_scratch = svt_lds + L.off_region; c.hme_scratch_bytes = (L.off_planes - L.off_region) + L.scratch_bytes; /* (me_lds_layout_geom: the planes follow the region) */
    c.quarter_sb  = svt_lds + L.off_quarter;
    c.ssdc        = pp.fractional_search_method == SVT_SSD_SEARCH ? (uint32_t *)(svt_lds + L.off_ssd) : nullptr;
    c.cand        = (uint32_t *)(svt_lds + L.off_cand);
    c.cand_hi     = (uint32_t *)(svt_lds + (L.off_cand_hi >= 0 ? L.off_cand_hi : L.off_cand));
    c.pred0       = (uint32_t *)(svt_lds + L.off_pred0);
    c.pic_w = pic_w; c.pic_h = pic_h; c.sb_index = sb; c.prof = prof;
    c.sb_x = (sb % nx) * ME_SB; c.sb_y = (sb / nx) * ME_SB;
    c.sb_w = (pic_w - c.sb_x) < ME_SB ? pic_w - c.sb_x : ME_SB;
    c.sb_h = (pic_h - c.sb_y) < ME_SB ? pic_h - c.sb_y : ME_SB;
    if constexpr (FAST) me_sb_run_fast<SPEC>(&c, threadIdx.x);
    else me_sb_run(&c, threadIdx.x);
}
#define ME_KERNEL_ATTRS(SPEC) __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(SPEC == 0 ? ME_WAVES_PER_EU : me_spec_waves_per_eu(SPEC) == 5 ? ME_WAVES_PER_EU_SPEC : me_spec_waves_per_eu(SPEC))))
template <int SPEC, bool COMPACT = false>
ME_KERNEL_ATTRS(SPEC) void svt_me_sb_kernel(const me_pic_dev *__restrict__ pics, svt_me_params p, me_lds_layout L,
                                                        int n_sb, int nx, int pic_w, int pic_h, int total, int chunk, unsigned long long *prof, uint32_t *redo) {
    me_kernel_body<SPEC, false, COMPACT>(pics, p, L, n_sb, nx, pic_w, pic_h, total, chunk, prof, redo);
}
/* the same for the presets me_fast.h serves (me_spec_fast), pictures of whole SB columns and level-0 areas up to 256 x 256 */
template <int SPEC>
#ifdef ME_FAST_NUM_VGPR
__attribute__((amdgpu_num_vgpr(ME_FAST_NUM_VGPR)))
#endif
ME_KERNEL_ATTRS(SPEC) void svt_me_fast_kernel(const me_pic_dev *__restrict__ pics, svt_me_params p, me_lds_layout L,
                                                          int n_sb, int nx, int pic_w, int pic_h, int total, int chunk, unsigned long long *prof) {
    me_kernel_body<SPEC, true>(pics, p, L, n_sb, nx, pic_w, pic_h, total, chunk, prof);
}

/* Stand-alone exhaustive SAD search = eb_vp9_sad_loop_kernel (C_DEFAULT/EbComputeSAD_C.c:132-169), one
 * workgroup per job; block and window are staged in LDS, the search is ph_sad_search. */
__global__ __launch_bounds__(256) void svt_sad_loop_kernel(const uint8_t *__restrict__ src, const uint8_t *__restrict__ ref,
                                                           const svt_sad_loop_job *__restrict__ jobs, int n_jobs,
                                                           svt_sad_loop_result *__restrict__ out, int lds_bytes) {
    const int j = blockIdx.x;
    if (j >= n_jobs) return;
    const svt_sad_loop_job job = jobs[j];
    const int              tid = threadIdx.x;
    me_ctx_t               c;
    me_state_t            &st_s = *(me_state_t *)svt_lds;
    c.st = &st_s;
    /* block rows: stride rounded to dwords */
    const int bstride = (job.width + 3) & ~3;
    uint8_t  *blk     = svt_lds + ((sizeof(me_state_t) + 15) & ~(size_t)15);
    uint8_t  *win     = blk + ((bstride * job.height + 15) & ~15);
    const int wbytes  = job.search_w + job.width + 3;
    int       wstride = ((wbytes + 3) & ~3) + 4;
    if (((wstride >> 2) & 1) == 0) wstride += 4;
    /* a search row y uses reference rows y*raw + j*ref_stride; the LDS window keeps rows at unit pitch
       `raw`, which requires ref_stride to be a multiple of ref_stride_raw (2 in every reference use) */
    const int mul      = job.ref_stride / job.ref_stride_raw;
    const int span     = mul * (job.height - 1);
    const int avail    = (lds_bytes - (int)(win - svt_lds)) / wstride;
    int       band     = avail - span;
    if (band > job.search_h) band = job.search_h;
    uint64_t  sad = 0xffffff;
    int       bx = 0, by = 0;
    ph_load_rect(tid, blk, bstride, src + job.src_off, job.src_stride, job.width, job.height);
    if (tid == 0) st_s.hme_key = ~0ull;
    __syncthreads();
    for (int y0 = 0; y0 < job.search_h && band > 0; y0 += band) {
        const int nr = y0 + band <= job.search_h ? band : job.search_h - y0;
        ph_load_rect(tid, win, wstride, ref + job.ref_off + (ptrdiff_t)y0 * job.ref_stride_raw, job.ref_stride_raw, wbytes, nr + span);
        __syncthreads();
        ph_sad_search(&c, tid, blk, bstride, job.width, job.height, win, wstride, job.search_w, nr, mul);
        __syncthreads();
        const uint64_t k = st_s.hme_key;
        if (k != ~0ull) {
            const uint32_t s = (uint32_t)(k >> 32), idx = (uint32_t)k + (uint32_t)(y0 * job.search_w);
            if (s < sad) { sad = s; bx = (int)(idx % (uint32_t)job.search_w); by = (int)(idx / (uint32_t)job.search_w); }
        }
        __syncthreads();
        if (tid == 0) st_s.hme_key = ~0ull;
        __syncthreads();
    }
    if (tid == 0) {
        out[j].best_sad = (uint32_t)sad;
        out[j].x = (int16_t)bx;
        out[j].y = (int16_t)by;
    }
}

/* ------------------------------------------------------------------------------------------------ */
/* launchers                                                                                          */
/* ------------------------------------------------------------------------------------------------ */
/* the fields that may differ between the pictures of one launch (me_spec.h: everything else is constant inside a configuration) */
static bool me_params_same_config(const svt_me_params *a, const svt_me_params *b) { return svt_hip_me_params_same_launch(a, b) != 0; }

/* params_stride = 0: one parameter set for every picture; 1: params[i] belongs to picture i */
static int32_t me_launch(svt_hip_ctx *ctx, int32_t n_pics, const svt_pa_picture *cur, const svt_pa_picture *ref0, const svt_pa_picture *ref1,
                         const svt_me_params *params_all, int params_stride, svt_me_pu_result *const *d_results, uint32_t *const *d_rcme) {
    if (!ctx || !cur || !ref0 || !params_all || !d_results || n_pics < 1) return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "me: null argument");
    const svt_me_params *params = params_all;
    for (int i = 0; i < (params_stride ? n_pics : 1); i++) {
        const svt_me_params *q = &params_all[i];
        if (q->num_ref_lists < 1 || q->num_ref_lists > 2) return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "me: num_ref_lists");
        if (q->num_ref_lists == 2 && !ref1) return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "me: ref1 missing for B picture");
        if (q->hierarchical_levels > 5 || q->temporal_layer_index > 5 || q->number_hme_search_region_in_width > 2 ||
            q->number_hme_search_region_in_height > 2)
            return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "me: parameter out of range");
        /* a search area of 0 x n has no position: the reference never produces one (verify_settings wants 1..256,
           Codec/EbEncHandle.c:2352-2360) and the search phases assume at least one */
        if (q->search_area_width < 1 || q->search_area_height < 1) return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "me: empty full-pel search area");
        if (i && !me_params_same_config(params, q))
            return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "me: the pictures of one launch may differ in num_ref_lists, temporal_layer_index, hierarchical_levels and same_ref_poc only");
    }
    me_lds_layout L;
    if (me_lds_layout_compute(params, &L)) return svt_set_error(SVT_HIP_ERR_UNSUPPORTED, "me: search area does not fit in LDS");
    const int W = cur[0].full.width, H = cur[0].full.height;
    if (W < 64 || H < 64 || (W & 7) || (H & 7)) return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "me: picture size");
    const int nx = (W + 63) / 64, ny = (H + 63) / 64, n_sb = nx * ny;
    HIP_TRY(hipSetDevice(ctx->device));
    /* picture descriptors -> device (pinned staging ring so that the copy is truly asynchronous) */
    me_pic_dev *h = nullptr, *d = nullptr;
    if (svt_ctx_stage(ctx, sizeof(me_pic_dev) * (size_t)n_pics, (void **)&h, (void **)&d)) return svt_set_error(SVT_HIP_ERR_NO_RESOURCES, "me: descriptor buffers");
    bool fast_ok = (W & 63) == 0; /* me_fast.h: whole SB columns, level-0 areas whose positions fit 8 bits */
    for (int i = 0; i < n_pics; i++) {
        if (cur[i].full.width != W || cur[i].full.height != H) return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "me: batch pictures differ in size");
        memset(&h[i], 0, sizeof h[i]);
        h[i].cur    = cur[i];
        h[i].ref[0] = ref0[i];
        if (ref1) h[i].ref[1] = ref1[i];
        h[i].results = d_results[i];
        h[i].rcme    = d_rcme ? d_rcme[i] : nullptr;
        const svt_me_params *q = &params_all[params_stride ? i : 0];
        me_lds_layout        Li;
        if (me_lds_layout_compute(q, &Li)) return svt_set_error(SVT_HIP_ERR_UNSUPPORTED, "me: search area does not fit in LDS");
        h[i].num_ref_lists = q->num_ref_lists; h[i].temporal_layer_index = q->temporal_layer_index;
        h[i].hierarchical_levels = q->hierarchical_levels; h[i].same_ref_poc = q->same_ref_poc;
        h[i].hme_w0[0] = Li.hme_w0[0]; h[i].hme_w0[1] = Li.hme_w0[1]; h[i].hme_h0[0] = Li.hme_h0[0]; h[i].hme_h0[1] = Li.hme_h0[1];
        h[i].hme_tw0 = Li.hme_tw0; h[i].hme_th0 = Li.hme_th0;
        {   /* me_fast.h: rows of the widest level-0 window (area + 16 x 8 block + alignment slack, odd dword stride) per scratch fill */
            const int wbytes = Li.hme_tw0 + 16 + 3;
            int       ws = ((wbytes + 3) & ~3) + 4;
            if (((ws >> 2) & 1) == 0) ws += 4;
            const int band = Li.scratch_bytes / ws - 14;
            h[i].hme_band = (int16_t)(band > 256 ? 256 : band);
            if (band < 1 || Li.hme_tw0 > 256 || Li.hme_th0 > 256) fast_ok = false;
        }
    }
    HIP_TRY(hipMemcpyAsync(d, h, sizeof(me_pic_dev) * (size_t)n_pics, hipMemcpyHostToDevice, ctx->stream));
    const int total = n_sb * n_pics, chunk = (n_sb + 7) / 8; /* SBs of one picture per XCD */
    /* SVT_HIP_ME_PROFILE=1: per-phase shader-cycle breakdown (thread 0 of every workgroup), printed to stderr */
    static const bool want_prof = getenv("SVT_HIP_ME_PROFILE") != nullptr;
    unsigned long long *d_prof = nullptr;
    if (want_prof) {
        d_prof = (unsigned long long *)svt_ctx_slot(ctx, 25, 32 * sizeof(unsigned long long));
        if (d_prof) HIP_TRY(hipMemsetAsync(d_prof, 0, 32 * sizeof(unsigned long long), ctx->stream));
    }
#ifdef ME_FINE_PROF
    { const char *sa = getenv("SVT_HIP_ME_STOP"); int v = sa ? atoi(sa) : -1; HIP_TRY(hipMemcpyToSymbolAsync(HIP_SYMBOL(g_me_stop_after), &v, sizeof v, 0, hipMemcpyHostToDevice, ctx->stream)); }
#endif
    /* compact layout: when the search area's width is a multiple of 8 and it buys a workgroup per CU (the 64 x 64-area presets) */
    static const bool no_compact = getenv("SVT_HIP_ME_NOCOMPACT") != nullptr;
    me_lds_layout Lc = L;
    uint32_t     *d_redo = nullptr;
    if (!no_compact && (params->search_area_width & 7) == 0 && params->search_area_width <= 127 && me_lds_layout_compute_ex(params, &Lc, 1) == 0 &&
        me_lds_workgroups_per_cu(&Lc) > me_lds_workgroups_per_cu(&L)) {
        d_redo = (uint32_t *)svt_ctx_slot(ctx, 39, (size_t)total * sizeof(uint32_t));
        if (!d_redo) return svt_set_error(SVT_HIP_ERR_NO_RESOURCES, "me: redo flags");
        HIP_TRY(hipMemsetAsync(d_redo, 0, (size_t)total * sizeof(uint32_t), ctx->stream));
    }
    HIP_TRY(hipEventRecord(ctx->ev_start, ctx->stream));
    /* the instance specialised for the caller's parameter set when there is one (me_spec.h), else the generic one */
    static const bool no_spec = getenv("SVT_HIP_ME_GENERIC") != nullptr;
    static const bool no_fast = getenv("SVT_HIP_ME_NOFAST") != nullptr;
    const int spec = no_spec ? 0 : me_spec_match(params);
    ctx->me_instance = spec + (d_redo ? 200 : 0);
    /* me_fast.h's driver where a compiled instance of it serves the parameter set (me_spec_fast; today SPEC 1) */
    bool fast_done = false;
#define ME_LAUNCH_FAST(S) \
    if constexpr (me_spec_fast(S)) { \
        if (L.total_bytes > 64 * 1024) \
            HIP_TRY(hipFuncSetAttribute((const void *)svt_me_fast_kernel<S>, hipFuncAttributeMaxDynamicSharedMemorySize, L.total_bytes)); \
        ctx->me_instance = 100 + S; \
        hipLaunchKernelGGL(svt_me_fast_kernel<S>, dim3(chunk * 8 * n_pics), dim3(256), L.total_bytes, ctx->stream, d, *params, L, n_sb, nx, W, H, total, chunk, d_prof); \
        fast_done = true; \
    }
    if (fast_ok && !no_fast && !d_redo)
        switch (spec) {
        case 1: ME_LAUNCH_FAST(1); break;
        default: break;
        }
#undef ME_LAUNCH_FAST
    if (!fast_done)
    switch (spec) {
#define ME_LAUNCH(S) \
    if (L.total_bytes > 64 * 1024) \
        HIP_TRY(hipFuncSetAttribute((const void *)svt_me_sb_kernel<S>, hipFuncAttributeMaxDynamicSharedMemorySize, L.total_bytes)); \
    hipLaunchKernelGGL(svt_me_sb_kernel<S>, dim3(chunk * 8 * n_pics), dim3(256), L.total_bytes, ctx->stream, d, *params, L, n_sb, nx, W, H, total, chunk, d_prof, (uint32_t *)nullptr)
    /* two launches where the compact layout (me_layout.h) buys a workgroup per CU: every SB with the compact layout, then the SBs that
     * flagged themselves (clipped search areas with tail columns: SBs near the right picture border) with the full one */
#define ME_LAUNCH2(S) \
    if (d_redo) { \
        if (Lc.total_bytes > 64 * 1024) HIP_TRY(hipFuncSetAttribute((const void *)svt_me_sb_kernel<S, true>, hipFuncAttributeMaxDynamicSharedMemorySize, Lc.total_bytes)); \
        if (L.total_bytes > 64 * 1024) HIP_TRY(hipFuncSetAttribute((const void *)svt_me_sb_kernel<S, false>, hipFuncAttributeMaxDynamicSharedMemorySize, L.total_bytes)); \
        hipLaunchKernelGGL((svt_me_sb_kernel<S, true>), dim3(chunk * 8 * n_pics), dim3(256), Lc.total_bytes, ctx->stream, d, *params, Lc, n_sb, nx, W, H, total, chunk, d_prof, d_redo); \
        hipLaunchKernelGGL((svt_me_sb_kernel<S, false>), dim3(chunk * 8 * n_pics), dim3(256), L.total_bytes, ctx->stream, d, *params, L, n_sb, nx, W, H, total, chunk, d_prof, d_redo); \
    } else { ME_LAUNCH(S); }
    case 1: ME_LAUNCH(1); break;
    case 2: ME_LAUNCH(2); break;
    case 3: ME_LAUNCH(3); break;
    case 4: ME_LAUNCH2(4); break;
    case 5: ME_LAUNCH2(5); break;
    default: ME_LAUNCH2(0); break;
#undef ME_LAUNCH2
#undef ME_LAUNCH
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(ctx->ev_stop, ctx->stream));
    if (d_prof) {
        unsigned long long hp[32];
        HIP_TRY(hipMemcpyAsync(hp, d_prof, sizeof hp, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        static const char *nm[14] = {"init", "center_sads", "hme", "zero_check", "region_load", "fullpel_sad8", "fullpel_argmin",
                                     "keys2best", "supel_enable", "interp", "halfpel", "quarterpel", "pred0_bipred", "output"};
        unsigned long long tot = 0;
        for (int i = 0; i < 14; i++) tot += hp[i];
        fprintf(stderr, "[me-profile] tl=%d pics=%d WGs=%d avg cycles/WG=%llu :", params->temporal_layer_index, n_pics, total, tot / (unsigned long long)total);
        for (int i = 0; i < 14; i++) fprintf(stderr, " %s=%.1f%%", nm[i], 100.0 * (double)hp[i] / (double)tot);
        fprintf(stderr, " | hme_load=%.1f%% hme_search=%.1f%%", 100.0 * (double)hp[14] / (double)tot, 100.0 * (double)hp[15] / (double)tot);
        if (hp[20] | hp[21]) fprintf(stderr, " hme_plan=%.1f%% hme_finish=%.1f%% hme_other=%.1f%%", 100.0 * (double)hp[20] / (double)tot, 100.0 * (double)hp[21] / (double)tot, 100.0 * (double)hp[22] / (double)tot);
        if (hp[16] | hp[17] | hp[18] | hp[19]) /* fine marks (builds with -DME_FINE_PROF): entry lookup, qsad block, key update, wave reductions */
            fprintf(stderr, " fine[lookup=%.1f%% qsad=%.1f%% keys=%.1f%% reduce=%.1f%%]", 100.0 * (double)hp[16] / (double)tot,
                    100.0 * (double)hp[17] / (double)tot, 100.0 * (double)hp[18] / (double)tot, 100.0 * (double)hp[19] / (double)tot);
        fprintf(stderr, "\n");
    }
    svt_ctx_stage_commit(ctx);
    ctx->timed = 1;
    return SVT_HIP_OK;
}

/* which compiled instance serves a parameter set: 0 = the generic one, 1..ME_SPEC_COUNT = a specialised one (me_spec.h) */
extern "C" int32_t svt_hip_me_kernel_instance(const svt_me_params *p) { return p ? me_spec_match(p) : -1; }
/* the instance the last ME launch on this context actually ran: the me_spec.h index, + 100 when it was me_fast.h's driver */
extern "C" int32_t svt_hip_me_last_instance(const svt_hip_ctx *ctx) { return ctx ? ctx->me_instance : -1; }
/* diagnostic (host only, no device needed): LDS bytes of a workgroup of the ME kernel for a parameter set, with the plain (compact = 0) or the
 * compact (1: search-area widths that are multiples of 8) layout of csrc/me_layout.h; a CU's 160 KB are handed out in granules of 1 280 bytes --
 * 32 000 bytes are five workgroups per CU, 81 920 two.  Negative: the layout does not exist. */
extern "C" int32_t svt_hip_me_lds_bytes(const svt_me_params *p, int32_t compact) {
    if (!p) return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "me_lds_bytes: null argument");
    if (compact && ((p->search_area_width & 7) != 0 || p->search_area_width > 127)) return -1;
    me_lds_layout L;
    if (me_lds_layout_compute_ex(p, &L, compact ? 1 : 0)) return -1;
    return L.total_bytes;
}

extern "C" int32_t svt_hip_me_batch_device(svt_hip_ctx *ctx, int32_t n_pics, const svt_pa_picture *cur,
                                           const svt_pa_picture *ref0, const svt_pa_picture *ref1,
                                           const svt_me_params *params, svt_me_pu_result *const *d_results,
                                           uint32_t *const *d_rcme) {
    return me_launch(ctx, n_pics, cur, ref0, ref1, params, 0, d_results, d_rcme);
}
extern "C" int32_t svt_hip_me_batch_layers_device(svt_hip_ctx *ctx, int32_t n_pics, const svt_pa_picture *cur,
                                                  const svt_pa_picture *ref0, const svt_pa_picture *ref1,
                                                  const svt_me_params *params, svt_me_pu_result *const *d_results,
                                                  uint32_t *const *d_rcme) {
    return me_launch(ctx, n_pics, cur, ref0, ref1, params, 1, d_results, d_rcme);
}

extern "C" int32_t svt_hip_me_picture_device(svt_hip_ctx *ctx, const svt_pa_picture *cur, const svt_pa_picture *ref0,
                                             const svt_pa_picture *ref1, const svt_me_params *params,
                                             svt_me_pu_result *d_results, uint32_t *d_rcme) {
    svt_me_pu_result *r[1] = {d_results};
    uint32_t         *c[1] = {d_rcme};
    return svt_hip_me_batch_device(ctx, 1, cur, ref0, ref1, params, r, d_rcme ? c : nullptr);
}

static int upload_plane(svt_hip_ctx *ctx, const svt_plane *h, svt_plane *d, int slot) {
    const size_t bytes = (size_t)h->stride * (size_t)(h->height + 2 * h->origin_y);
    uint8_t     *dev   = (uint8_t *)svt_ctx_slot(ctx, slot, bytes + 64);
    if (!dev) return -1;
    if (hipMemcpyAsync(dev, h->buf, bytes, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) return -1;
    *d     = *h;
    d->buf = dev;
    return 0;
}

extern "C" int32_t svt_hip_me_picture(svt_hip_ctx *ctx, const svt_pa_picture *cur, const svt_pa_picture *ref0,
                                      const svt_pa_picture *ref1, const svt_me_params *params, svt_me_pu_result *results,
                                      uint32_t *rcme) {
    if (!ctx || !cur || !ref0 || !params || !results) return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "me: null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    svt_pa_picture        dc, d0, d1;
    const svt_pa_picture *hp[3] = {cur, ref0, ref1};
    svt_pa_picture       *dp[3] = {&dc, &d0, &d1};
    memset(&d1, 0, sizeof d1);
    for (int i = 0; i < 3; i++) {
        if (!hp[i]) continue;
        const int need_q = params->enable_hme_level_1_flag, need_s = params->enable_hme_level_0_flag;
        *dp[i] = *hp[i];
        if (upload_plane(ctx, &hp[i]->full, &dp[i]->full, 3 * i)) return svt_set_error(SVT_HIP_ERR_DEVICE, "me: upload");
        if (need_q && upload_plane(ctx, &hp[i]->quarter, &dp[i]->quarter, 3 * i + 1)) return svt_set_error(SVT_HIP_ERR_DEVICE, "me: upload");
        if (need_s && upload_plane(ctx, &hp[i]->sixteenth, &dp[i]->sixteenth, 3 * i + 2)) return svt_set_error(SVT_HIP_ERR_DEVICE, "me: upload");
    }
    const int         n_sb = svt_hip_sb_count(cur->full.width, cur->full.height);
    svt_me_pu_result *dr   = (svt_me_pu_result *)svt_ctx_slot(ctx, 9, sizeof(svt_me_pu_result) * 85 * (size_t)n_sb);
    uint32_t         *drc  = rcme ? (uint32_t *)svt_ctx_slot(ctx, 10, sizeof(uint32_t) * (size_t)n_sb) : nullptr;
    if (!dr || (rcme && !drc)) return svt_set_error(SVT_HIP_ERR_NO_RESOURCES, "me: result buffers");
    int32_t rc = svt_hip_me_picture_device(ctx, &dc, &d0, ref1 ? &d1 : nullptr, params, dr, drc);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(results, dr, sizeof(svt_me_pu_result) * 85 * (size_t)n_sb, hipMemcpyDeviceToHost, ctx->stream));
    if (rcme) HIP_TRY(hipMemcpyAsync(rcme, drc, sizeof(uint32_t) * (size_t)n_sb, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return SVT_HIP_OK;
}

/* compute_zz_sad (Codec/EbMotionEstimationProcess.c:431-534): one wave per SB, lane = 4 samples of the 16x16 block */
__global__ __launch_bounds__(256) void svt_me_zz_sad_kernel(svt_plane cur16, svt_plane prev, int nx, int n_sb, int shift,
                                                            uint32_t *__restrict__ zz, uint8_t *__restrict__ nmi) {
    const int sb = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (sb >= n_sb) return;
    const int sx = (sb % nx) * ME_SB, sy = (sb / nx) * ME_SB;
    const int complete = sx + ME_SB <= prev.width && sy + ME_SB <= prev.height;
    uint32_t  sad = 0;
    if (complete) {
        const int r = lane >> 2, c4 = (lane & 3) * 4; /* row of the 16x16 block, first of 4 columns */
        const uint8_t *a = cur16.buf + (size_t)(cur16.origin_y + (sy >> 2) + r) * cur16.stride + cur16.origin_x + (sx >> 2) + c4;
        const uint8_t *b = prev.buf + (size_t)(prev.origin_y + sy + 4 * r) * prev.stride + prev.origin_x + sx + 4 * c4;
        _Pragma("unroll") for (int k = 0; k < 4; k++) { const int d = (int)a[k] - (int)b[4 * k]; sad += (uint32_t)(d < 0 ? -d : d); }
    }
    svt_wave_add_u32_to_lane0(&sad);
    if (lane == 0) {
        const uint32_t v = complete ? sad : 0xffffffffu;
        zz[sb] = v;
        const uint32_t base = 16 * 16; /* block_width * block_height of a complete SB's 1/16 block */
        nmi[sb] = v < ((base * 2) >> shift) ? 0 : v < ((base * 4) >> shift) ? 10 : v < ((base * 8) >> shift) ? 20 : 30;
    }
}

extern "C" int32_t svt_hip_me_zz_sad_device(svt_hip_ctx *ctx, const svt_plane *cur_sixteenth, const svt_plane *prev_input,
                                            int32_t input_resolution, uint32_t *d_zz_sad, uint8_t *d_non_moving_index) {
    if (!ctx || !cur_sixteenth || !prev_input || !d_zz_sad || !d_non_moving_index || input_resolution < 0 || input_resolution > 3)
        return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "zz_sad: bad argument");
    static const int th_shift[4] = {4, 2, 0, 0}; /* non_moving_th_shift, Codec/EbMotionEstimationProcess.c:353 */
    HIP_TRY(hipSetDevice(ctx->device));
    const int nx = (prev_input->width + ME_SB - 1) / ME_SB, ny = (prev_input->height + ME_SB - 1) / ME_SB, n_sb = nx * ny;
    HIP_TRY(hipEventRecord(ctx->ev_start, ctx->stream));
    hipLaunchKernelGGL(svt_me_zz_sad_kernel, dim3((n_sb + 3) / 4), dim3(256), 0, ctx->stream, *cur_sixteenth, *prev_input, nx, n_sb,
                       th_shift[input_resolution], d_zz_sad, d_non_moving_index);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(ctx->ev_stop, ctx->stream));
    ctx->timed = 1;
    return SVT_HIP_OK;
}

extern "C" int32_t svt_hip_sad_loop_batch_device(svt_hip_ctx *ctx, const uint8_t *d_src, const uint8_t *d_ref,
                                                 const svt_sad_loop_job *d_jobs, int32_t n_jobs, svt_sad_loop_result *d_out) {
    if (!ctx || !d_src || !d_ref || !d_jobs || !d_out || n_jobs < 1) return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "sad_loop: null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    const int lds = 64 * 1024;
    HIP_TRY(hipEventRecord(ctx->ev_start, ctx->stream));
    hipLaunchKernelGGL(svt_sad_loop_kernel, dim3(n_jobs), dim3(256), lds, ctx->stream, d_src, d_ref, d_jobs, n_jobs, d_out, lds);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(ctx->ev_stop, ctx->stream));
    ctx->timed = 1;
    return SVT_HIP_OK;
}


/* The rest of the ME kernel process's per-SB bookkeeping (row M12): stationary_edge_over_update_over_time_sb_part1 / _part2
 * (Codec/EbMotionEstimationProcess.c:785-869) and the rate-control SAD-interval indices / histograms (:1103-1237), one thread
 * per SB straight from the ME results, the picture-analysis variances and the rcme distortions in HBM.  The histograms are
 * first gathered per workgroup in LDS (128 + 128 bins), then added to the picture's. */
__device__ __forceinline__ uint32_t me_sad_interval(uint32_t v) { /* :1126-1137 */
    uint32_t i = (v & 0xffffu) >> 2;
    if (i > 63) i = 63 + ((i - 63) >> 3);
    return i >= 127 ? 127 : i;
}
__global__ __launch_bounds__(256) void svt_me_sb_stats_kernel(svt_me_sb_stats_params p, int nx, int n_sb, const svt_me_pu_result *__restrict__ results,
                                                              const uint16_t *__restrict__ var, const uint32_t *__restrict__ rcme,
                                                              svt_me_sb_stats *__restrict__ out, uint32_t *__restrict__ hist, uint32_t *__restrict__ full_count) {
    __shared__ uint32_t s_hist[2 * SVT_SAD_INTERVALS + 1];
    for (int i = threadIdx.x; i < 2 * SVT_SAD_INTERVALS + 1; i += 256) s_hist[i] = 0;
    __syncthreads();
    const int sb = blockIdx.x * 256 + threadIdx.x;
    if (sb < n_sb) {
        const int W = p.pic_width, H = p.pic_height, ox = (sb % nx) * ME_SB, oy = (sb / nx) * ME_SB;
        const int complete = ox + ME_SB <= W && oy + ME_SB <= H;
        /* potential_logo_sb (Codec/EbSequenceControlSet.c:332-413): the top corners and the bottom band, 3 x 2 / 7 x 4 / 14 x 8 SBs */
        const int k = p.input_resolution <= 0 ? 1 : p.input_resolution < 3 ? 2 : 4;
        const int wx = (k == 1 ? 3 : 7 * (k >> 1)) * ME_SB, wy = 2 * k * ME_SB;
        const int logo = complete && ((oy < wy && (ox >= W - wx || ox < wx)) || oy >= H - wy);
        uint32_t  check1 = 0, pm1 = 0, check2 = 0, low = 0, inter_idx = 0, intra_idx = 0;
        if (logo) {
            int mvx = 0, mvy = 0;
            if (p.temporal_layer_index > 0 && results) { mvx = results[(size_t)sb * 85].x_mv_l0; mvy = results[(size_t)sb * 85].y_mv_l0; }
            const bool     low_motion = p.temporal_layer_index == 0 || ((mvx < 0 ? -mvx : mvx) < 16 && (mvy < 0 ? -mvy : mvy) < 16);
            const uint16_t *v = var + (size_t)sb * 85;
            const int64_t  v0 = v[1], v1 = v[2], v2 = v[3], v3 = v[4], avg = (v0 + v1 + v2 + v3) >> 2;
            const int32_t  d0 = (int32_t)(v0 - avg), d1 = (int32_t)(v1 - avg), d2 = (int32_t)(v2 - avg), d3 = (int32_t)(v3 - avg);
            /* int32 products that wrap, an arithmetic shift of the int32 sum, then the widening (:806-811) */
            const int32_t  s4 = (int32_t)((uint32_t)d0 * (uint32_t)d0 + (uint32_t)d1 * (uint32_t)d1 + (uint32_t)d2 * (uint32_t)d2 + (uint32_t)d3 * (uint32_t)d3);
            const uint64_t vov = (uint64_t)(int64_t)(s4 >> 2);
            check1 = !(vov <= 50000 || !low_motion);
            pm1    = vov > 1000;
        }
        if (p.run_part2) {
            if (logo) {
                const uint32_t th = p.input_resolution < 2 ? 5u : 2u;
                const uint32_t dist = (p.slice_type == 0 && results) ? results[(size_t)sb * 85].distortion_direction[0].distortion : 0u;
                low = p.slice_type == 0 && dist < 64u * 64u * th;
            }
            check2 = 1; /* [quirk] :868 */
        }
        if (p.rate_control_mode && complete) {
            if (p.slice_type != 2) {
                inter_idx = me_sad_interval(rcme[sb] >> 8);
                atomicAdd(&s_hist[inter_idx], 1u);
            }
            intra_idx = me_sad_interval((uint32_t)var[(size_t)sb * 85] >> 4);
            atomicAdd(&s_hist[SVT_SAD_INTERVALS + intra_idx], 1u);
            atomicAdd(&s_hist[2 * SVT_SAD_INTERVALS], 1u);
        }
        svt_me_sb_stats o;
        o.check1_for_logo_stationary_edge_over_time_flag = (uint8_t)check1; o.pm_check1_for_logo_stationary_edge_over_time_flag = (uint8_t)pm1;
        o.check2_for_logo_stationary_edge_over_time_flag = (uint8_t)check2; o.low_dist_logo = (uint8_t)low;
        o.inter_sad_interval_index = (uint16_t)inter_idx; o.intra_sad_interval_index = (uint16_t)intra_idx;
        out[sb] = o;
    }
    __syncthreads();
    if (p.rate_control_mode) {
        for (int i = threadIdx.x; i < 2 * SVT_SAD_INTERVALS; i += 256) if (s_hist[i]) atomicAdd(&hist[i], s_hist[i]);
        if (threadIdx.x == 0 && s_hist[2 * SVT_SAD_INTERVALS]) atomicAdd(full_count, s_hist[2 * SVT_SAD_INTERVALS]);
    }
}

extern "C" int32_t svt_hip_me_sb_stats_device(svt_hip_ctx *ctx, const svt_me_sb_stats_params *params, const svt_me_pu_result *d_results,
                                              const uint16_t *d_var, const uint32_t *d_rcme, svt_me_sb_stats *d_stats, uint32_t *d_hist,
                                              uint32_t *d_full_sb_count) {
    if (!ctx || !params || !d_var || !d_stats) return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "sb_stats: null argument");
    if (params->pic_width < 8 || params->pic_height < 8 || params->input_resolution < 0 || params->input_resolution > 3 ||
        params->slice_type < 0 || params->slice_type > 2)
        return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "sb_stats: parameter out of range");
    if (params->slice_type != 2 && !d_results) return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "sb_stats: results missing for an inter picture");
    if (params->rate_control_mode && (!d_hist || !d_full_sb_count || (params->slice_type != 2 && !d_rcme)))
        return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "sb_stats: histogram / rcme buffers missing");
    HIP_TRY(hipSetDevice(ctx->device));
    const int nx = (params->pic_width + ME_SB - 1) / ME_SB, ny = (params->pic_height + ME_SB - 1) / ME_SB, n_sb = nx * ny;
    HIP_TRY(hipEventRecord(ctx->ev_start, ctx->stream));
    hipLaunchKernelGGL(svt_me_sb_stats_kernel, dim3((n_sb + 255) / 256), dim3(256), 0, ctx->stream, *params, nx, n_sb, d_results, d_var, d_rcme, d_stats,
                       d_hist, d_full_sb_count);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(ctx->ev_stop, ctx->stream));
    ctx->timed = 1;
    return SVT_HIP_OK;
}
