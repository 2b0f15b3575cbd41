/*
 * encdec.hip -- picture-level EncDec driver (gfx950): everything the encode pass does with mode decision's output, for a batch of
 * mutually independent pictures, device resident from the mode-info grid to the padded reference picture.
 *
 * Replaces the data path of eb_vp9_enc_dec_kernel behind mode decision (Source/Lib/Codec/EbEncDecProcess.c:5306):
 *   encode_pass_sb (:3627-4241): inter prediction (:3787-3802)  -> svt_mc_kernel
 *                                perform_coding_loop per block   -> svt_tq_kernel<N> over lists built here from the grid
 *                                skip flags (:4069-4098)         -> svt_tq_skip_kernel + svt_skip_update_kernel
 *   last SB of the picture (:5625-5697): eb_vp9_build_mask_frame -> svt_lf_mask_kernel (device form of svt_hip_lf_build_masks)
 *                                eb_vp9_loop_filter_frame        -> svt_lf_kernel
 *                                pad_ref_and_set_flags           -> svt_refpad_kernel
 * The reference walks SBs and blocks one at a time on the host; here the host only enqueues: the block lists, skip flags and
 * filter masks are derived on the device from the grid (rules: encdec_core.h, shared with the host forms the CPU tests pin), so a
 * picture costs no per-block descriptor traffic over PCIe and no host wait between the stages.
 *
 * List building is three small kernels: count (one wave per SB, one lane per 8x8 unit; per-SB totals per transform size), an
 * exclusive scan over [size][picture][SB] (one workgroup), emit (same mapping; a lane's place inside its SB comes from a wave
 * prefix sum).  The order is deterministic: size, picture, SB raster, unit raster, luma / Cb / Cr -- SB-ordered inside a size, so
 * the blocks a wave of the transform kernel takes lie next to each other in the planes.  These kernels move a few MB per picture
 * and are bound by launch latency, not by bandwidth; they exist to keep the chain on the device.
 */
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>
#include "svt_ctx.h"
#include "encdec_core.h"

#define ED_MAX_PICS 32
#define ED_MAX_SETS 8 /* reconstruction base pointers a transform launch can address (svt_tq_block.pad_[0] bits 4-6) */

struct svt_encdec_work {
    int           max_pics, width, height, n_sb, sb_cols, mi_rows, mi_cols;
    size_t        cap_per_pic;   /* transform blocks of a picture at most: all 4x4 */
    svt_tq_block *d_blocks;      /* only when a host asks for the descriptor list (svt_hip_encdec_work_download): allocated and filled then */
    void         *last_hb;       /* host copy of the latest batch's parameter block (for that) */
    uint32_t     *d_pos;
    uint16_t     *d_eob;
    int32_t      *d_counts;      /* [4][max_pics][n_sb], turned into offsets by the scan */
    int32_t      *d_totals, *d_bases; /* [4][max_pics]: blocks of a (size, picture) / its first block */
    int32_t      *d_off_cnt;     /* [8]: first block / number of blocks per size */
    int32_t      *d_status;      /* != 0: a malformed grid was seen */
    int32_t      *d_intra_sync;  /* ticket and per-(plane, 16x16 cell) flags of the intra kernel: 2 + 3 * (at most 16 per SB) dwords */
    svt_quant_tables *d_qtabs;   /* [2] luma, chroma of the batch's q index */
    int16_t      *d_iscan;
    int           last_pics;
    /* SB-ordered lists (round 6, the default): [picture][chunk of SVT_TQ_CHUNK_SBS SBs of an SB row][size][SB][unit][plane]; d_counts then holds,
       per picture, n_chunks x 4 x SVT_TQ_CHUNK_SBS per-SB counts turned into exclusive prefixes, d_bases n_pics + 1 first blocks */
    int           sb_order, chunks_per_row, n_chunks;
    svt_encdec_stage_hook hook;  /* profiling aid: called on the enqueueing thread at every stage boundary */
    void         *hook_user;
};

namespace {

struct ed_pic_dev {
    const svt_me_pu_result *res;    /* stand-in decision only */
    svt_mc_mode_info       *mc_mi;
    svt_lf_mode_info       *lf_mi;
    uint8_t                *nz;
    uint16_t               *eob_map;
    svt_lf_mask            *lfm;
    svt_tq_pic_geom         g;
};
struct ed_batch_dev {
    ed_pic_dev pic[ED_MAX_PICS];
    uint32_t   iscan_off[16];
    int32_t    n_pics, n_sb, sb_cols, mi_rows, mi_cols, mi_stride, width, height;
    uint32_t   lambda;
    int32_t    filter_level;
};

__device__ __forceinline__ int wave_sum(int v) {
    _Pragma("unroll") for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
__device__ __forceinline__ int wave_excl_prefix(int v, int lane) {
    int incl = v;
    _Pragma("unroll") for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(incl, d, 64);
        if (lane >= d) incl += o;
    }
    return incl - v;
}

/* one wave per (picture, SB), one lane per 8x8 unit: per-SB number of transform blocks of each size; also clears the unit's
 * "has coefficients" scratch byte for the skip pass */
__global__ __launch_bounds__(64) void svt_tq_count_kernel(const ed_batch_dev *__restrict__ B, int32_t *__restrict__ counts, int32_t *__restrict__ status) {
    const int sb = (int)blockIdx.x % B->n_sb, pic = (int)blockIdx.x / B->n_sb, lane = (int)threadIdx.x;
    const ed_pic_dev &P = B->pic[pic];
    const int ur = (sb / B->sb_cols) * 8 + (lane >> 3), uc = (sb % B->sb_cols) * 8 + (lane & 7);
    int cnt[4] = {0, 0, 0, 0};
    if (ur < B->mi_rows && uc < B->mi_cols) P.nz[ur * B->mi_stride + uc] = 0;
    const int o = svt_tq_unit_is_origin(P.lf_mi, B->mi_stride, B->mi_rows, B->mi_cols, ur, uc);
    if (o < 0) atomicOr(status, 1);
    if (o == 1) svt_tq_unit_counts(P.lf_mi, B->mi_stride, ur, uc, cnt);
    _Pragma("unroll") for (int s = 0; s < 4; s++) {
        const int v = wave_sum(cnt[s]);
        if (lane == 0) counts[(s * B->n_pics + pic) * B->n_sb + sb] = v;
    }
}

/* Offsets of the lists, two small kernels instead of one long serial one.  Level 1: one workgroup per (size, picture) turns the
 * picture's per-SB counts into exclusive prefixes in place and leaves the picture's total in totals[size * n_pics + picture].  Level 2:
 * one wave turns the totals into the first block of every (size, picture) (bases[], in list order: size, then picture) and into
 * off_cnt[s] / off_cnt[4 + s] = first block / number of blocks of size s.  The emit kernel adds bases[] to the per-SB prefixes. */
__global__ __launch_bounds__(256) void svt_scan_sb_kernel(int32_t *__restrict__ a, int n_sb, int32_t *__restrict__ totals) {
    __shared__ int32_t part[256];
    int32_t *p = a + (size_t)blockIdx.x * n_sb;
    const int t = (int)threadIdx.x, per = (n_sb + 255) / 256, b = t * per, e = b + per < n_sb ? b + per : n_sb;
    int s = 0;
    for (int i = b; i < e; i++) s += p[i];
    part[t] = s;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        const int v = t >= d ? part[t - d] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int run = part[t] - s;
    for (int i = b; i < e; i++) { const int v = p[i]; p[i] = run; run += v; }
    if (t == 255) totals[blockIdx.x] = part[255];
}
__global__ __launch_bounds__(64) void svt_scan_pic_kernel(const int32_t *__restrict__ totals, int n_pics, int32_t *__restrict__ bases, int32_t *__restrict__ off_cnt) {
    if (threadIdx.x != 0) return; /* 4 x n_pics (<= 128) values: not worth more than one lane */
    int run = 0;
    for (int s = 0; s < 4; s++) {
        off_cnt[s] = run;
        for (int p = 0; p < n_pics; p++) { bases[s * n_pics + p] = run; run += totals[s * n_pics + p]; }
        off_cnt[4 + s] = run - off_cnt[s];
    }
}

/* the descriptor list a host may ask for (svt_hip_encdec_work_download): rebuilt from the position codes, the transform size of the range a
 * block sits in and the pictures' geometry -- the same function the transform kernels call per block */
__global__ __launch_bounds__(256) void svt_tq_expand_kernel(const ed_batch_dev *__restrict__ B, const int32_t *__restrict__ off_cnt, const uint32_t *__restrict__ pos,
                                                           svt_tq_block *__restrict__ blocks) {
    const int i = (int)blockIdx.x * 256 + (int)threadIdx.x, total = off_cnt[3] + off_cnt[7];
    if (i >= total) return;
    const int      ts = i >= off_cnt[3] ? 3 : i >= off_cnt[2] ? 2 : i >= off_cnt[1] ? 1 : 0;
    const uint32_t p = pos[i];
    svt_tq_block   k;
    svt_tq_block_from_pos(p, ts, &B->pic[svt_tq_pos_pic(p)].g, B->iscan_off, B->sb_cols, &k);
    blocks[i] = k;
}

/* same mapping as the count kernel: every block-origin lane writes its descriptors at offsets[size][picture][SB] + (blocks of that
 * size of the lanes before it) */
__global__ __launch_bounds__(64) void svt_tq_emit_kernel(const ed_batch_dev *__restrict__ B, const int32_t *__restrict__ offsets, const int32_t *__restrict__ bases,
                                                         svt_tq_block *__restrict__ blocks, uint32_t *__restrict__ pos) {
    const int sb = (int)blockIdx.x % B->n_sb, pic = (int)blockIdx.x / B->n_sb, lane = (int)threadIdx.x;
    const ed_pic_dev &P = B->pic[pic];
    const int ur = (sb / B->sb_cols) * 8 + (lane >> 3), uc = (sb % B->sb_cols) * 8 + (lane & 7);
    int cnt[4] = {0, 0, 0, 0};
    const int o = svt_tq_unit_is_origin(P.lf_mi, B->mi_stride, B->mi_rows, B->mi_cols, ur, uc);
    if (o == 1) svt_tq_unit_counts(P.lf_mi, B->mi_stride, ur, uc, cnt);
    uint32_t base[4];
    _Pragma("unroll") for (int s = 0; s < 4; s++) base[s] = (uint32_t)(bases[s * B->n_pics + pic] + offsets[(s * B->n_pics + pic) * B->n_sb + sb] + wave_excl_prefix(cnt[s], lane));
    if (o == 1) svt_tq_unit_emit(P.lf_mi, B->mi_stride, ur, uc, &P.g, B->iscan_off, base, blocks, pos);
}

/* ---- SB-ordered lists (one transform launch, tq_kernel.hip: svt_tq_sb_kernel) ----
 * counts2[pic][chunk][size][k]: transform blocks of size `size` in SB k of the chunk; the exclusive prefix over a picture's row of
 * n_chunks * 4 * CH entries IS the list order [chunk][size][SB], so one scan gives every SB's slot and every (chunk, size) segment. */
__device__ __forceinline__ int ed_seg_index(const ed_batch_dev *B, int chunks_per_row, int pic, int sb, int s) {
    const int sr = sb / B->sb_cols, sc = sb - sr * B->sb_cols, chunk = sr * chunks_per_row + sc / SVT_TQ_CHUNK_SBS, k = sc % SVT_TQ_CHUNK_SBS;
    const int n_chunks = chunks_per_row * ((B->height + 63) >> 6);
    return ((pic * n_chunks + chunk) * 4 + s) * SVT_TQ_CHUNK_SBS + k;
}
__global__ __launch_bounds__(64) void svt_tq_count_sb_kernel(const ed_batch_dev *__restrict__ B, int32_t *__restrict__ counts, int32_t *__restrict__ status, int chunks_per_row) {
    const int sb = (int)blockIdx.x % B->n_sb, pic = (int)blockIdx.x / B->n_sb, lane = (int)threadIdx.x;
    const ed_pic_dev &P = B->pic[pic];
    const int ur = (sb / B->sb_cols) * 8 + (lane >> 3), uc = (sb % B->sb_cols) * 8 + (lane & 7);
    int cnt[4] = {0, 0, 0, 0};
    if (ur < B->mi_rows && uc < B->mi_cols) P.nz[ur * B->mi_stride + uc] = 0;
    const int o = svt_tq_unit_is_origin(P.lf_mi, B->mi_stride, B->mi_rows, B->mi_cols, ur, uc);
    if (o < 0) atomicOr(status, 1);
    if (o == 1) svt_tq_unit_counts(P.lf_mi, B->mi_stride, ur, uc, cnt);
    _Pragma("unroll") for (int s = 0; s < 4; s++) {
        const int v = wave_sum(cnt[s]);
        if (lane == 0) counts[ed_seg_index(B, chunks_per_row, pic, sb, s)] = v;
    }
}
/* one workgroup per picture: exclusive prefix over the picture's row in place, the picture's total, and its blocks per size */
__global__ __launch_bounds__(256) void svt_scan_seg_kernel(int32_t *__restrict__ a, int n_per_pic, int32_t *__restrict__ totals, int32_t *__restrict__ size_tot /* [4], zeroed */) {
    __shared__ int32_t part[256];
    __shared__ int32_t szs[4];
    int32_t *p = a + (size_t)blockIdx.x * n_per_pic;
    const int t = (int)threadIdx.x, per = (n_per_pic + 255) / 256, b = t * per, e = b + per < n_per_pic ? b + per : n_per_pic;
    if (t < 4) szs[t] = 0;
    int s = 0, sz[4] = {0, 0, 0, 0};
    for (int i = b; i < e; i++) { const int v = p[i]; s += v; sz[(i / SVT_TQ_CHUNK_SBS) & 3] += v; }
    part[t] = s;
    __syncthreads();
    _Pragma("unroll") for (int k = 0; k < 4; k++) { const int v = wave_sum(sz[k]); if ((t & 63) == 0 && v) atomicAdd(&szs[k], v); }
    for (int d = 1; d < 256; d <<= 1) {
        const int v = t >= d ? part[t - d] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int run = part[t] - s;
    for (int i = b; i < e; i++) { const int v = p[i]; p[i] = run; run += v; }
    if (t == 255) totals[blockIdx.x] = part[255];
    if (t < 4 && szs[t]) atomicAdd(&size_tot[t], szs[t]);
}
/* bases[pic] (n_pics + 1 entries: the last = the batch's total) and off_cnt in the layout every consumer of the totals knows: [s] = blocks of the
 * smaller sizes, [4 + s] = blocks of size s (so off_cnt[3] + off_cnt[7] = total, as with the size-grouped lists) */
__global__ __launch_bounds__(64) void svt_scan_pic_sb_kernel(const int32_t *__restrict__ totals, int n_pics, int32_t *__restrict__ bases, int32_t *__restrict__ off_cnt,
                                                             const int32_t *__restrict__ size_tot) {
    if (threadIdx.x != 0) return;
    int run = 0;
    for (int p = 0; p < n_pics; p++) { bases[p] = run; run += totals[p]; }
    bases[n_pics] = run;
    int o = 0;
    for (int s = 0; s < 4; s++) { off_cnt[s] = o; off_cnt[4 + s] = size_tot[s]; o += size_tot[s]; }
}
__global__ __launch_bounds__(64) void svt_tq_emit_sb_kernel(const ed_batch_dev *__restrict__ B, const int32_t *__restrict__ seg, const int32_t *__restrict__ bases,
                                                            uint32_t *__restrict__ pos, int chunks_per_row) {
    const int sb = (int)blockIdx.x % B->n_sb, pic = (int)blockIdx.x / B->n_sb, lane = (int)threadIdx.x;
    const ed_pic_dev &P = B->pic[pic];
    const int ur = (sb / B->sb_cols) * 8 + (lane >> 3), uc = (sb % B->sb_cols) * 8 + (lane & 7);
    int cnt[4] = {0, 0, 0, 0};
    const int o = svt_tq_unit_is_origin(P.lf_mi, B->mi_stride, B->mi_rows, B->mi_cols, ur, uc);
    if (o == 1) svt_tq_unit_counts(P.lf_mi, B->mi_stride, ur, uc, cnt);
    uint32_t base[4];
    _Pragma("unroll") for (int s = 0; s < 4; s++) base[s] = (uint32_t)(bases[pic] + seg[ed_seg_index(B, chunks_per_row, pic, sb, s)] + wave_excl_prefix(cnt[s], lane));
    if (o == 1) svt_tq_unit_emit(P.lf_mi, B->mi_stride, ur, uc, &P.g, B->iscan_off, base, (svt_tq_block *)nullptr, pos);
}
/* descriptor list for a host that asks (svt_hip_encdec_work_download), SB-ordered lists: one workgroup per (picture, chunk, size) segment */
__global__ __launch_bounds__(64) void svt_tq_expand_sb_kernel(const ed_batch_dev *__restrict__ B, const int32_t *__restrict__ seg, const int32_t *__restrict__ bases,
                                                              const uint32_t *__restrict__ pos, svt_tq_block *__restrict__ blocks, int n_chunks) {
    const int s = (int)blockIdx.x & 3, item = (int)blockIdx.x >> 2, pic = item / n_chunks, chunk = item - pic * n_chunks;
    const int per_pic = n_chunks * 4 * SVT_TQ_CHUNK_SBS, at = (chunk * 4 + s) * SVT_TQ_CHUNK_SBS, nxt = at + SVT_TQ_CHUNK_SBS;
    const int first = bases[pic] + seg[pic * per_pic + at], last = nxt < per_pic ? bases[pic] + seg[pic * per_pic + nxt] : bases[pic + 1];
    for (int i = first + (int)threadIdx.x; i < last; i += 64) {
        svt_tq_block k;
        svt_tq_block_from_pos(pos[i], s, &B->pic[svt_tq_pos_pic(pos[i])].g, B->iscan_off, B->sb_cols, &k);
        blocks[i] = k;
    }
}

/* after the transform stage: the eob of every block goes to its place in the picture's eob map, and a block with coefficients marks
 * the prediction block it belongs to (the skip flag of the reference is "no coefficient in any transform block of the block",
 * Codec/EbEncDecProcess.c:4069-4098; the four 4x4 quadrants of a sub-8x8 unit share one flag the same way) */
__global__ __launch_bounds__(256) void svt_tq_skip_kernel(const ed_batch_dev *__restrict__ B, const int32_t *__restrict__ off_cnt, const uint32_t *__restrict__ pos,
                                                          const uint16_t *__restrict__ eob) {
    const int total = off_cnt[3] + off_cnt[7];
    const int w4 = B->width >> 2, h4 = B->height >> 2;
    for (int i = (int)(blockIdx.x * blockDim.x + threadIdx.x); i < total; i += (int)(gridDim.x * blockDim.x)) {
        const uint32_t p = pos[i];
        const int      pic = (int)(p >> 24) & 63, plane = (int)(p >> 22) & 3, y4 = (int)(p >> 11) & 0x7ff, x4 = (int)p & 0x7ff;
        const ed_pic_dev &P = B->pic[pic];
        const int      e = eob[i];
        const int      pw4 = plane ? w4 >> 1 : w4;
        const int      off = plane == 0 ? 0 : w4 * h4 + (plane == 2 ? (w4 >> 1) * (h4 >> 1) : 0);
        P.eob_map[off + y4 * pw4 + x4] = (uint16_t)e;
        if (e) {
            const int uy = plane ? y4 : y4 >> 1, ux = plane ? x4 : x4 >> 1;
            const svt_lf_mode_info b = P.lf_mi[uy * B->mi_stride + ux];
            const int w8 = svt_blk_w8(b.sb_type), h8 = svt_blk_h8(b.sb_type);
            P.nz[(uy - uy % h8) * B->mi_stride + (ux - ux % w8)] = 1;
        }
    }
}
/* the eob maps of the batch read 0 before the blocks write theirs: one launch for all pictures (16 bytes per lane) */
__global__ __launch_bounds__(256) void svt_eob_map_clear_kernel(const ed_batch_dev *__restrict__ B, int n16) {
    uint4 *m = (uint4 *)B->pic[blockIdx.y].eob_map;
    for (int i = (int)(blockIdx.x * 256 + threadIdx.x); i < n16; i += (int)(gridDim.x * 256)) m[i] = make_uint4(0, 0, 0, 0);
}
/* every unit takes the skip flag of its block */
__global__ __launch_bounds__(256) void svt_skip_update_kernel(const ed_batch_dev *__restrict__ B) {
    const int units = B->mi_rows * B->mi_cols;
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= units * B->n_pics) return;
    const int pic = i / units, u = i % units, ur = u / B->mi_cols, uc = u % B->mi_cols;
    const ed_pic_dev &P = B->pic[pic];
    svt_lf_mode_info *b = &P.lf_mi[ur * B->mi_stride + uc];
    if (b->sb_type > 12) return;
    const int w8 = svt_blk_w8(b->sb_type), h8 = svt_blk_h8(b->sb_type);
    b->skip = P.nz[(ur - ur % h8) * B->mi_stride + (uc - uc % w8)] ? 0 : 1;
}

/* one wave per (picture, SB), one lane per 8x8 unit: the unit's contribution (svt_lf_mask_unit, the text svt_hip_lf_build_masks runs
 * on the host) and a wave-wide OR; lane u writes lfl_y[u], lanes 0..18 the mask words */
__device__ __forceinline__ uint32_t wave_or(uint32_t v) {
    _Pragma("unroll") for (int d = 32; d >= 1; d >>= 1) v |= (uint32_t)__shfl_xor((int)v, d, 64);
    return v;
}
__global__ __launch_bounds__(64) void svt_lf_mask_kernel(const ed_batch_dev *__restrict__ B, int32_t *__restrict__ status) {
    const int sb = (int)blockIdx.x % B->n_sb, pic = (int)blockIdx.x / B->n_sb, lane = (int)threadIdx.x;
    const ed_pic_dev &P = B->pic[pic];
    svt_lf_unit_masks u;
    svt_lf_mask_unit(P.lf_mi, B->mi_stride, B->mi_rows, B->mi_cols, sb / B->sb_cols, sb % B->sb_cols, lane >> 3, lane & 7, &u);
    if (u.bad && status) atomicOr(status, 1);
    /* the 160-byte mask is assembled in LDS and leaves as 40 consecutive dwords (one coalesced store instead of a dozen partial ones) */
    __shared__ uint32_t sm[40];
    uint64_t            w64[9];
    _Pragma("unroll") for (int i = 0; i < 4; i++) { w64[i] = u.left_y[i]; w64[4 + i] = u.above_y[i]; }
    w64[8] = u.int_4x4_y;
    _Pragma("unroll") for (int i = 0; i < 9; i++) {   /* left_y[4], above_y[4], int_4x4_y: the first nine quadwords of the mask */
        const uint32_t lo = wave_or((uint32_t)w64[i]), hi = wave_or((uint32_t)(w64[i] >> 32));
        if (lane == i) { sm[2 * i] = lo; sm[2 * i + 1] = hi; }
    }
    /* the nine 16-bit chroma words, two per dword: left_uv[0..3], above_uv[0..3], int_4x4_uv (bytes 72..89) */
    const uint32_t c0 = wave_or((uint32_t)u.left_uv[0] | (uint32_t)u.left_uv[1] << 16), c1 = wave_or((uint32_t)u.left_uv[2] | (uint32_t)u.left_uv[3] << 16);
    const uint32_t c2 = wave_or((uint32_t)u.above_uv[0] | (uint32_t)u.above_uv[1] << 16), c3 = wave_or((uint32_t)u.above_uv[2] | (uint32_t)u.above_uv[3] << 16);
    const uint32_t c4 = wave_or((uint32_t)u.int_4x4_uv);
    if (lane == 9) { sm[18] = c0; sm[19] = c1; sm[20] = c2; sm[21] = c3; sm[22] = c4; sm[38] = 0; sm[39] = 0; } /* (dword 22's upper half and dwords 38-39 are rewritten below) */
    __syncthreads();
    ((uint8_t *)sm)[90 + lane] = u.level;   /* lfl_y[64]: bytes 90..153 */
    __syncthreads();
    if (lane < 40) ((uint32_t *)&P.lfm[sb])[lane] = sm[lane];
}

/* stand-in decision: one wave per (picture, SB), one lane per unit */
__global__ __launch_bounds__(64) void svt_md_default_kernel(const ed_batch_dev *__restrict__ B) {
    const int sb = (int)blockIdx.x % B->n_sb, pic = (int)blockIdx.x / B->n_sb, lane = (int)threadIdx.x;
    const ed_pic_dev &P = B->pic[pic];
    const int r = lane >> 3, c = lane & 7, sr = sb / B->sb_cols, sc = sb % B->sb_cols, ur = sr * 8 + r, uc = sc * 8 + c;
    if (ur >= B->mi_rows || uc >= B->mi_cols) return;
    svt_md_default_unit(P.res + (size_t)sb * SVT_ME_PU_COUNT, r, c, sr, sc, B->mi_rows, B->mi_cols, B->lambda, B->filter_level, &P.mc_mi[ur * B->mi_stride + uc],
                        &P.lf_mi[ur * B->mi_stride + uc]);
}

/* the batch's parameter block goes to the device through the context's descriptor ring; the slot is committed at once: every
 * consumer is enqueued on the same stream behind the copy, and a later re-use of the slot is a copy on that stream too */
int stage_batch(svt_hip_ctx *ctx, const ed_batch_dev &hb, const ed_batch_dev **d_out) {
    void *h = nullptr, *d = nullptr;
    if (svt_ctx_stage(ctx, sizeof hb, &h, &d)) return -1;
    memcpy(h, &hb, sizeof hb);
    if (hipMemcpyAsync(d, h, sizeof hb, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) return -1;
    svt_ctx_stage_commit(ctx);
    *d_out = (const ed_batch_dev *)d;
    return 0;
}

void fill_dims(ed_batch_dev &hb, int n_pics, int width, int height, int mi_stride) {
    hb.n_pics = n_pics; hb.width = width; hb.height = height;
    hb.sb_cols = (width + 63) >> 6;
    hb.n_sb = hb.sb_cols * ((height + 63) >> 6);
    hb.mi_rows = height >> 3; hb.mi_cols = width >> 3; hb.mi_stride = mi_stride;
}

} // namespace

extern "C" int32_t svt_hip_encdec_work_create(svt_hip_ctx *ctx, int32_t max_pics, int32_t width, int32_t height, svt_encdec_work **out) {
    if (!ctx || !out || max_pics < 1 || max_pics > ED_MAX_PICS || width < 8 || height < 8 || (width & 7) || (height & 7))
        return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "encdec_work: bad argument");
    HIP_TRY(hipSetDevice(ctx->device));
    svt_encdec_work *w = (svt_encdec_work *)calloc(1, sizeof *w);
    if (!w) return svt_set_error(SVT_HIP_ERR_NO_RESOURCES, "encdec_work: malloc");
    w->max_pics = max_pics; w->width = width; w->height = height;
    w->sb_cols = (width + 63) >> 6; w->n_sb = w->sb_cols * ((height + 63) >> 6);
    w->mi_rows = height >> 3; w->mi_cols = width >> 3;
    w->cap_per_pic = (size_t)width * height * 3 / 32;
    /* SVT_HIP_TQ_SB_ORDER=1: SB-ordered lists and the SB-ordered transform launch(es) (svt_tq_sb_kernel) instead of the size-grouped lists with one
     * launch per transform size.  Measured on MI355X, 2160p, 16 pictures (profiles/r06_pmc_traffic.md): the stage's HBM-side traffic drops from
     * 2.08 GB to 1.50 GB (reads 1.19 -> 0.55 GB: an SB is fetched once instead of once per size) -- and the step gets SLOWER (10.2 -> 10.5 ms; the
     * stage alone 0.63 -> 0.68 ms): the one launch holds the 32x32 body's 128 registers and 16.9 KB of LDS in every workgroup and balances worse
     * than the persistent size-grouped walks; the bytes it saves were not what bounded the stage.  So the size-grouped form stays the default;
     * read per workspace so that the tests cover both. */
    { const char *e = getenv("SVT_HIP_TQ_SB_ORDER"); w->sb_order = e && atoi(e) != 0; }
    w->chunks_per_row = (w->sb_cols + SVT_TQ_CHUNK_SBS - 1) / SVT_TQ_CHUNK_SBS;
    w->n_chunks = w->chunks_per_row * ((height + 63) >> 6);
    const size_t cap = w->cap_per_pic * (size_t)max_pics;
    const size_t n_cnt = (size_t)4 * max_pics * (w->sb_order ? (size_t)w->n_chunks * SVT_TQ_CHUNK_SBS : (size_t)w->n_sb);
    uint32_t const *offs = nullptr;
    int32_t         entries = 0;
    const int16_t  *isc = svt_hip_vp9_iscan_tables(&offs, &entries);
    /* the lists are position codes (4 bytes per transform block; the 32-byte descriptors are rebuilt in registers where they are used) */
    bool ok = hipMalloc((void **)&w->d_pos, cap * sizeof(uint32_t)) == hipSuccess &&
              hipMalloc((void **)&w->d_eob, cap * sizeof(uint16_t)) == hipSuccess &&
              hipMalloc((void **)&w->d_counts, (n_cnt + 16 + 8 * ED_MAX_PICS + 16 + 48 * (size_t)w->n_sb) * sizeof(int32_t)) == hipSuccess &&
              hipMalloc((void **)&w->d_qtabs, 2 * sizeof(svt_quant_tables)) == hipSuccess && hipMalloc((void **)&w->d_iscan, (size_t)entries * sizeof(int16_t)) == hipSuccess;
    if (ok) {
        w->d_off_cnt = w->d_counts + n_cnt;
        w->d_status = w->d_off_cnt + 8;   /* (d_off_cnt[9 .. 12]: per-size totals of the SB-ordered scan) */
        w->d_totals = w->d_off_cnt + 16;
        w->d_bases = w->d_totals + 4 * ED_MAX_PICS;   /* 4 * ED_MAX_PICS entries (size-grouped) / n_pics + 1 (SB-ordered) */
        w->d_intra_sync = w->d_bases + 4 * ED_MAX_PICS + 4;
        ok = hipMemsetAsync(w->d_off_cnt, 0, 16 * sizeof(int32_t), ctx->stream) == hipSuccess &&
             hipMemcpyAsync(w->d_iscan, isc, (size_t)entries * sizeof(int16_t), hipMemcpyHostToDevice, ctx->stream) == hipSuccess &&
             hipStreamSynchronize(ctx->stream) == hipSuccess;
    }
    if (!ok) {
        svt_hip_encdec_work_destroy(ctx, w);
        return svt_set_error(SVT_HIP_ERR_NO_RESOURCES, "encdec_work: device memory");
    }
    *out = w;
    return SVT_HIP_OK;
}

extern "C" void svt_hip_encdec_work_destroy(svt_hip_ctx *ctx, svt_encdec_work *w) {
    if (!w) return;
    if (ctx) { (void)hipSetDevice(ctx->device); (void)hipStreamSynchronize(ctx->stream); }
    void *v[6] = {w->d_blocks, w->d_pos, w->d_eob, w->d_counts, w->d_qtabs, w->d_iscan};
    for (int i = 0; i < 6; i++) if (v[i]) (void)hipFree(v[i]);
    free(w->last_hb);
    free(w);
}

extern "C" int32_t svt_hip_encdec_batch_device(svt_hip_ctx *ctx, svt_encdec_work *w, int32_t n_pics, const svt_encdec_picture *pics, int32_t width,
                                               int32_t height, int32_t mi_stride, int32_t q_index, const svt_encdec_flags *flags, const svt_lf_thresh *thr,
                                               int32_t pad_x, int32_t pad_y) {
    if (!ctx || !w || !pics || !flags || n_pics < 1 || n_pics > w->max_pics || width != w->width || height != w->height || mi_stride < (width >> 3))
        return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "encdec: bad argument");
    if (flags->apply_loop_filter && !thr) return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "encdec: loop filter without thresholds");
    HIP_TRY(hipSetDevice(ctx->device));
    /* one 32-bit offset space for the source planes and one for the prediction planes of the batch */
    uintptr_t src_lo = UINTPTR_MAX, src_hi = 0, pred_lo = UINTPTR_MAX, pred_hi = 0;
    for (int i = 0; i < n_pics; i++) {
        const svt_encdec_picture &p = pics[i];
        if (!p.d_mc_mi || !p.d_lf_mi || !p.src.y || !p.src.u || !p.src.v || !p.pred.y || !p.pred.u || !p.pred.v || !p.recon.y || !p.recon.u || !p.recon.v ||
            !p.d_qcoeff || !p.d_eob_map || !p.d_nz || (flags->apply_loop_filter && !p.d_lfm))
            return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "encdec: null picture field");
        if (!p.d_dqcoeff != !pics[0].d_dqcoeff) return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "encdec: d_dqcoeff must be given for every picture of a batch or for none");
        if (((uintptr_t)p.d_qcoeff | (uintptr_t)p.d_dqcoeff) & 15) return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "encdec: coefficient arrays must be 16-byte aligned");
        const uint8_t *s3[3] = {p.src.y, p.src.u, p.src.v}, *p3[3] = {p.pred.y, p.pred.u, p.pred.v};
        for (int k = 0; k < 3; k++) {
            src_lo = (uintptr_t)s3[k] < src_lo ? (uintptr_t)s3[k] : src_lo; src_hi = (uintptr_t)s3[k] > src_hi ? (uintptr_t)s3[k] : src_hi;
            pred_lo = (uintptr_t)p3[k] < pred_lo ? (uintptr_t)p3[k] : pred_lo; pred_hi = (uintptr_t)p3[k] > pred_hi ? (uintptr_t)p3[k] : pred_hi;
        }
    }
    /* bytes from a plane's first sample to its last row's end: the largest stride x rows among the planes the 32-bit block offsets
       address (a reference buffer's stride is width + 160; a caller's stride can exceed 2 x width on narrow pictures) */
    uint64_t plane_span = 0;
    for (int i = 0; i < n_pics; i++) {
        const svt_yuv_planes *pl[3] = {&pics[i].src, &pics[i].pred, &pics[i].recon};
        for (int k = 0; k < 3; k++) {
            const uint64_t a = (uint64_t)(pl[k]->y_stride > 0 ? pl[k]->y_stride : 0) * (uint64_t)height, b = (uint64_t)(pl[k]->uv_stride > 0 ? pl[k]->uv_stride : 0) * (uint64_t)(height / 2);
            plane_span = a > plane_span ? a : plane_span;
            plane_span = b > plane_span ? b : plane_span;
        }
    }
    if ((uint64_t)(src_hi - src_lo) + plane_span >= (1ull << 32) || (uint64_t)(pred_hi - pred_lo) + plane_span >= (1ull << 32))
        return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "encdec: the batch's source / prediction planes must lie within 4 GB");
    /* coefficient arrays: one base for the batch as well (element offsets are 32 bit) */
    uintptr_t q_lo = UINTPTR_MAX, dq_lo = UINTPTR_MAX;
    for (int i = 0; i < n_pics; i++) {
        q_lo = (uintptr_t)pics[i].d_qcoeff < q_lo ? (uintptr_t)pics[i].d_qcoeff : q_lo;
        dq_lo = (uintptr_t)pics[i].d_dqcoeff < dq_lo ? (uintptr_t)pics[i].d_dqcoeff : dq_lo;
    }
    ed_batch_dev hb;
    memset(&hb, 0, sizeof hb);
    fill_dims(hb, n_pics, width, height, mi_stride);
    {
        const uint32_t *offs = nullptr;
        (void)svt_hip_vp9_iscan_tables(&offs, nullptr);
        for (int i = 0; i < 16; i++) hb.iscan_off[i] = offs[i];
    }
    /* reconstruction bases: at most ED_MAX_SETS pointers, each serving every picture whose planes lie within 4 GB above it.  Chosen
       from the pictures' lowest plane addresses in ascending order, so that the result does not depend on the order the pictures
       arrive in (separately allocated buffers in descending address order once opened a base each) */
    uint8_t       *recon_set[ED_MAX_SETS];
    int            n_sets = 0;
    {
        uintptr_t lo[ED_MAX_PICS], hi[ED_MAX_PICS];
        int       order[ED_MAX_PICS];
        for (int i = 0; i < n_pics; i++) {
            const uint8_t *r3[3] = {pics[i].recon.y, pics[i].recon.u, pics[i].recon.v};
            lo[i] = UINTPTR_MAX; hi[i] = 0; order[i] = i;
            for (int k = 0; k < 3; k++) { lo[i] = (uintptr_t)r3[k] < lo[i] ? (uintptr_t)r3[k] : lo[i]; hi[i] = (uintptr_t)r3[k] > hi[i] ? (uintptr_t)r3[k] : hi[i]; }
        }
        for (int a = 1; a < n_pics; a++) /* insertion sort by lowest address (n_pics <= 32) */
            for (int b = a; b > 0 && lo[order[b]] < lo[order[b - 1]]; b--) { const int t_ = order[b]; order[b] = order[b - 1]; order[b - 1] = t_; }
        for (int a = 0; a < n_pics; a++) {
            const int i = order[a];
            if (n_sets && (uint64_t)(hi[i] - (uintptr_t)recon_set[n_sets - 1]) + plane_span < (1ull << 32)) continue; /* (sorted: lo[i] >= the last base) */
            if (n_sets == ED_MAX_SETS || (uint64_t)(hi[i] - lo[i]) + plane_span >= (1ull << 32))
                return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "encdec: the batch's reconstruction buffers span more than 8 regions of 4 GB");
            recon_set[n_sets++] = (uint8_t *)lo[i];
        }
    }
    svt_mc_picture mcp[ED_MAX_PICS];
    svt_yuv_planes rec[ED_MAX_PICS], rec_pad[ED_MAX_PICS];
    int            n_rec_pad = 0;
    const svt_lf_mask *lfm[ED_MAX_PICS];
    int32_t        lfm_stride[ED_MAX_PICS], mi_rows_a[ED_MAX_PICS], mi_cols_a[ED_MAX_PICS];
    for (int i = 0; i < n_pics; i++) {
        const svt_encdec_picture &p = pics[i];
        ed_pic_dev &P = hb.pic[i];
        P.mc_mi = (svt_mc_mode_info *)p.d_mc_mi; P.lf_mi = p.d_lf_mi; P.nz = p.d_nz; P.eob_map = p.d_eob_map; P.lfm = p.d_lfm;
        const uintptr_t q_off = ((uintptr_t)p.d_qcoeff - q_lo) / sizeof(int16_t), dq_off = ((uintptr_t)p.d_dqcoeff - dq_lo) / sizeof(int16_t);
        if ((p.d_dqcoeff && q_off != dq_off) || q_off + (uint64_t)hb.n_sb * SVT_SB_COEFFS >= (1ull << 32))
            return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "encdec: qcoeff / dqcoeff of the batch must be laid out alike, within 2^32 elements");
        /* the picture's reconstruction planes: 32-bit offsets from one of at most ED_MAX_SETS base pointers (a base serves every
           picture whose planes lie within 4 GB above it: buffers carved out of one slab share one) */
        const uint8_t *r3[3] = {p.recon.y, p.recon.u, p.recon.v};
        uintptr_t r_lo = UINTPTR_MAX, r_hi = 0;
        for (int k = 0; k < 3; k++) { r_lo = (uintptr_t)r3[k] < r_lo ? (uintptr_t)r3[k] : r_lo; r_hi = (uintptr_t)r3[k] > r_hi ? (uintptr_t)r3[k] : r_hi; }
        int set = -1;
        for (int k = n_sets - 1; k >= 0 && set < 0; k--) /* the highest base at or below the picture: the one chosen for it above */
            if (r_lo >= (uintptr_t)recon_set[k] && (uint64_t)(r_hi - (uintptr_t)recon_set[k]) + plane_span < (1ull << 32)) set = k;
        if (set < 0) return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "encdec: the batch's reconstruction buffers span more than 8 regions of 4 GB");
        svt_tq_pic_geom &g = P.g;
        const uint8_t *s3[3] = {p.src.y, p.src.u, p.src.v}, *p3[3] = {p.pred.y, p.pred.u, p.pred.v};
        for (int k = 0; k < 3; k++) {
            g.src_off[k] = (uint32_t)((uintptr_t)s3[k] - src_lo); g.pred_off[k] = (uint32_t)((uintptr_t)p3[k] - pred_lo);
            g.recon_off[k] = (uint32_t)((uintptr_t)r3[k] - (uintptr_t)recon_set[set]);
        }
        if (p.src.y_stride > 65535 || p.pred.y_stride > 65535 || p.recon.y_stride > 65535) return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "encdec: stride");
        g.src_stride[0] = (uint16_t)p.src.y_stride; g.src_stride[1] = (uint16_t)p.src.uv_stride;
        g.pred_stride[0] = (uint16_t)p.pred.y_stride; g.pred_stride[1] = (uint16_t)p.pred.uv_stride;
        g.recon_stride[0] = (uint16_t)p.recon.y_stride; g.recon_stride[1] = (uint16_t)p.recon.uv_stride;
        g.coeff_base = (uint32_t)q_off; g.width = width; g.height = height; g.recon_set = (uint8_t)set; g.pic = (uint8_t)i; g.do_recon = flags->do_recon ? 1 : 0;
        mcp[i].d_mi = p.d_mc_mi; mcp[i].mi_stride = mi_stride; mcp[i].mi_rows = hb.mi_rows; mcp[i].mi_cols = hb.mi_cols;
        mcp[i].ref[0] = p.ref[0]; mcp[i].ref[1] = p.ref[1]; mcp[i].pred = p.pred; mcp[i].use_subpel = p.use_subpel;
        rec[i] = p.recon; rec[i].width = width; rec[i].height = height;
        if (!p.no_pad) rec_pad[n_rec_pad++] = rec[i];
        lfm[i] = p.d_lfm; lfm_stride[i] = hb.sb_cols; mi_rows_a[i] = hb.mi_rows; mi_cols_a[i] = hb.mi_cols;
    }
    const ed_batch_dev *dB = nullptr;
    if (stage_batch(ctx, hb, &dB)) return svt_set_error(SVT_HIP_ERR_NO_RESOURCES, "encdec: descriptor buffers");
    /* quantiser tables of the q index */
    {
        svt_quant_tables qt[2];
        if (svt_hip_quant_tables_for_qindex(q_index, qt) != SVT_HIP_OK) return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "encdec: q index");
        void *h = nullptr, *d = nullptr;
        if (svt_ctx_stage(ctx, sizeof qt, &h, &d)) return svt_set_error(SVT_HIP_ERR_NO_RESOURCES, "encdec: descriptor buffers");
        memcpy(h, qt, sizeof qt);
        HIP_TRY(hipMemcpyAsync(w->d_qtabs, h, sizeof qt, hipMemcpyHostToDevice, ctx->stream));
        svt_ctx_stage_commit(ctx);
    }
#define ED_STAGE(k) do { if (w->hook) w->hook(w->hook_user, (k)); } while (0)
    /* 1. inter prediction of the whole batch */
    ED_STAGE(SVT_ENCDEC_STAGE_MC);
    int32_t rc = svt_hip_inter_pred_batch_device(ctx, n_pics, mcp);
    if (rc) return rc;
    /* 2. transform blocks from the grids */
    ED_STAGE(SVT_ENCDEC_STAGE_LISTS);
    const int nwg = n_pics * hb.n_sb;
    const int seg_per_pic = w->n_chunks * 4 * SVT_TQ_CHUNK_SBS;
    if (w->sb_order) {
        /* (a chunk at the right edge may hold fewer SBs than SVT_TQ_CHUNK_SBS: its missing entries count 0) */
        if (w->sb_cols % SVT_TQ_CHUNK_SBS) HIP_TRY(hipMemsetAsync(w->d_counts, 0, (size_t)n_pics * seg_per_pic * sizeof(int32_t), ctx->stream));
        HIP_TRY(hipMemsetAsync(w->d_off_cnt + 9, 0, 4 * sizeof(int32_t), ctx->stream));
        hipLaunchKernelGGL(svt_tq_count_sb_kernel, dim3(nwg), dim3(64), 0, ctx->stream, dB, w->d_counts, w->d_status, w->chunks_per_row);
        hipLaunchKernelGGL(svt_scan_seg_kernel, dim3(n_pics), dim3(256), 0, ctx->stream, w->d_counts, seg_per_pic, w->d_totals, w->d_off_cnt + 9);
        hipLaunchKernelGGL(svt_scan_pic_sb_kernel, dim3(1), dim3(64), 0, ctx->stream, (const int32_t *)w->d_totals, n_pics, w->d_bases, w->d_off_cnt, (const int32_t *)(w->d_off_cnt + 9));
        hipLaunchKernelGGL(svt_tq_emit_sb_kernel, dim3(nwg), dim3(64), 0, ctx->stream, dB, (const int32_t *)w->d_counts, (const int32_t *)w->d_bases, w->d_pos, w->chunks_per_row);
    } else {
    hipLaunchKernelGGL(svt_tq_count_kernel, dim3(nwg), dim3(64), 0, ctx->stream, dB, w->d_counts, w->d_status);
    hipLaunchKernelGGL(svt_scan_sb_kernel, dim3(4 * n_pics), dim3(256), 0, ctx->stream, w->d_counts, hb.n_sb, w->d_totals);
    hipLaunchKernelGGL(svt_scan_pic_kernel, dim3(1), dim3(64), 0, ctx->stream, (const int32_t *)w->d_totals, n_pics, w->d_bases, w->d_off_cnt);
    hipLaunchKernelGGL(svt_tq_emit_kernel, dim3(nwg), dim3(64), 0, ctx->stream, dB, (const int32_t *)w->d_counts, (const int32_t *)w->d_bases, (svt_tq_block *)nullptr, w->d_pos);
    }
    HIP_TRY(hipGetLastError());
    if (!w->last_hb) w->last_hb = malloc(sizeof hb);
    if (w->last_hb) memcpy(w->last_hb, &hb, sizeof hb);
    /* 3. residual -> transform -> quantisation (-> inverse -> reconstruction) */
    ED_STAGE(SVT_ENCDEC_STAGE_TQ);
    int32_t cap[4];
    for (int s = 0; s < 4; s++) cap[s] = (int32_t)((size_t)n_pics * width * height * 3 / 2 / (size_t)(16 << (2 * s)));
    if (w->sb_order)
        rc = svt_tq_launch_sb_lists(ctx, (const uint8_t *)src_lo, (const uint8_t *)pred_lo, recon_set, n_sets, w->d_qtabs, w->d_iscan, (int16_t *)q_lo, (int16_t *)dq_lo, w->d_eob,
                                    w->d_pos, &dB->pic[0].g, (int)sizeof(ed_pic_dev), dB->iscan_off, hb.sb_cols, w->d_counts, w->d_bases, n_pics, w->n_chunks, 4 * SVT_TQ_CHUNK_SBS);
    else
    rc = svt_tq_launch_device_lists(ctx, (const uint8_t *)src_lo, (const uint8_t *)pred_lo, recon_set, n_sets, nullptr, cap, w->d_off_cnt, w->d_qtabs, w->d_iscan,
                                    (int16_t *)q_lo, (int16_t *)dq_lo, w->d_eob, nullptr, w->d_pos, &dB->pic[0].g, (int)sizeof(ed_pic_dev), dB->iscan_off, hb.sb_cols);
    if (rc) return rc;
    /* 4. eob map + skip flags (entries of the map that are not the origin of a transform block of THIS picture read 0) */
    ED_STAGE(SVT_ENCDEC_STAGE_SKIP);
    {
        const size_t map_bytes = (size_t)(width / 4) * (height / 4) * 3 / 2 * sizeof(uint16_t);
        uintptr_t    low = map_bytes;
        for (int i = 0; i < n_pics; i++) low |= (uintptr_t)pics[i].d_eob_map;
        if (!(low & 15))
            hipLaunchKernelGGL(svt_eob_map_clear_kernel, dim3(64, n_pics), dim3(256), 0, ctx->stream, dB, (int)(map_bytes / 16));
        else
            for (int i = 0; i < n_pics; i++) HIP_TRY(hipMemsetAsync(pics[i].d_eob_map, 0, map_bytes, ctx->stream));
    }
    /* 3b. intra blocks of inter pictures: their inter neighbours are reconstructed now; the wavefront kernel codes them, one picture
           after the other (it writes their eob-map entries and coefficient flags itself) */
    for (int i = 0; i < n_pics; i++)
        if (pics[i].has_intra) {
            if (!flags->do_recon) return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "encdec: intra blocks need the reconstruction (do_recon)");
            rc = svt_intra_launch(ctx, &pics[i], width, height, mi_stride, w->d_qtabs, w->d_iscan, hb.iscan_off, w->d_intra_sync, w->d_status, 1);
            if (rc) return rc;
        }
    {
        const int total_cap = (int)(w->cap_per_pic * (size_t)n_pics);
        int       g = (total_cap + 255) / 256;
        if (g > ctx->cu_count * 8) g = ctx->cu_count * 8;
        hipLaunchKernelGGL(svt_tq_skip_kernel, dim3(g), dim3(256), 0, ctx->stream, dB, (const int32_t *)w->d_off_cnt, (const uint32_t *)w->d_pos, (const uint16_t *)w->d_eob);
        hipLaunchKernelGGL(svt_skip_update_kernel, dim3((n_pics * hb.mi_rows * hb.mi_cols + 255) / 256), dim3(256), 0, ctx->stream, dB);
        HIP_TRY(hipGetLastError());
    }
    /* 5. deblocking */
    ED_STAGE(SVT_ENCDEC_STAGE_LF);
    if (flags->apply_loop_filter) {
        hipLaunchKernelGGL(svt_lf_mask_kernel, dim3(nwg), dim3(64), 0, ctx->stream, dB, w->d_status);
        HIP_TRY(hipGetLastError());
        rc = svt_hip_lf_batch_device(ctx, n_pics, rec, lfm, lfm_stride, thr, mi_rows_a, mi_cols_a, 0);
        if (rc) return rc;
    }
    /* 6. the reconstruction becomes a reference picture */
    ED_STAGE(SVT_ENCDEC_STAGE_PAD);
    if (flags->pad_reference && n_rec_pad) {
        rc = svt_hip_ref_pad_batch_device(ctx, n_rec_pad, rec_pad, pad_x, pad_y);
        if (rc) return rc;
    }
    ED_STAGE(SVT_ENCDEC_STAGE_END);
    w->last_pics = n_pics;
    return SVT_HIP_OK;
}

/* An intra picture: prediction + transform + reconstruction by the wavefront kernel (intra_kernel.hip), then the same tail as a batch of
 * inter pictures -- skip flags, loop-filter masks, deblocking, border. */
extern "C" int32_t svt_hip_encdec_intra_device(svt_hip_ctx *ctx, svt_encdec_work *w, const svt_encdec_picture *pic, int32_t width, int32_t height, int32_t mi_stride,
                                               int32_t q_index, const svt_encdec_flags *flags, const svt_lf_thresh *thr, int32_t pad_x, int32_t pad_y) {
    if (!ctx || !w || !pic || !flags || width != w->width || height != w->height || mi_stride < (width >> 3))
        return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "encdec_intra: bad argument");
    if (!flags->do_recon) return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "encdec_intra: intra prediction needs the reconstruction (do_recon)");
    if (flags->apply_loop_filter && !thr) return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "encdec_intra: loop filter without thresholds");
    const svt_encdec_picture &p = *pic;
    if (!p.d_lf_mi || !p.src.y || !p.src.u || !p.src.v || !p.recon.y || !p.recon.u || !p.recon.v || !p.d_qcoeff || !p.d_eob_map || !p.d_nz ||
        (flags->apply_loop_filter && !p.d_lfm))
        return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "encdec_intra: null picture field");
    if (((uintptr_t)p.d_qcoeff | (uintptr_t)p.d_dqcoeff) & 15) return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "encdec_intra: coefficient arrays must be 16-byte aligned");
    if (p.recon.y_stride > 65535 || ((p.src.y_stride | p.src.uv_stride | p.recon.y_stride | p.recon.uv_stride) & 3) ||
        (((uintptr_t)p.src.y | (uintptr_t)p.src.u | (uintptr_t)p.src.v | (uintptr_t)p.recon.y | (uintptr_t)p.recon.u | (uintptr_t)p.recon.v) & 3) ||
        (p.pred.y && (((uintptr_t)p.pred.y | (uintptr_t)p.pred.u | (uintptr_t)p.pred.v | (uintptr_t)p.pred.y_stride | (uintptr_t)p.pred.uv_stride) & 3)))
        return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "encdec_intra: planes and strides must be 4-byte aligned");
    if (p.pred.y && (!p.pred.u || !p.pred.v)) return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "encdec_intra: prediction planes");
    HIP_TRY(hipSetDevice(ctx->device));
    ed_batch_dev hb;
    memset(&hb, 0, sizeof hb);
    fill_dims(hb, 1, width, height, mi_stride);
    const uint32_t *offs = nullptr;
    (void)svt_hip_vp9_iscan_tables(&offs, nullptr);
    hb.pic[0].lf_mi = p.d_lf_mi; hb.pic[0].nz = p.d_nz; hb.pic[0].eob_map = p.d_eob_map; hb.pic[0].lfm = p.d_lfm;
    const ed_batch_dev *dB = nullptr;
    if (stage_batch(ctx, hb, &dB)) return svt_set_error(SVT_HIP_ERR_NO_RESOURCES, "encdec_intra: descriptor buffers");
    {
        svt_quant_tables qt[2];
        if (svt_hip_quant_tables_for_qindex(q_index, qt) != SVT_HIP_OK) return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "encdec_intra: q index");
        void *h = nullptr, *d = nullptr;
        if (svt_ctx_stage(ctx, sizeof qt, &h, &d)) return svt_set_error(SVT_HIP_ERR_NO_RESOURCES, "encdec_intra: descriptor buffers");
        memcpy(h, qt, sizeof qt);
        HIP_TRY(hipMemcpyAsync(w->d_qtabs, h, sizeof qt, hipMemcpyHostToDevice, ctx->stream));
        svt_ctx_stage_commit(ctx);
    }
    ED_STAGE(SVT_ENCDEC_STAGE_TQ);
    HIP_TRY(hipMemsetAsync(p.d_eob_map, 0, (size_t)(width / 4) * (height / 4) * 3 / 2 * sizeof(uint16_t), ctx->stream));
    HIP_TRY(hipMemsetAsync(p.d_nz, 0, (size_t)mi_stride * hb.mi_rows, ctx->stream));
    int32_t rc = svt_intra_launch(ctx, pic, width, height, mi_stride, w->d_qtabs, w->d_iscan, offs, w->d_intra_sync, w->d_status, 0);
    if (rc) return rc;
    ED_STAGE(SVT_ENCDEC_STAGE_SKIP);
    hipLaunchKernelGGL(svt_skip_update_kernel, dim3((hb.mi_rows * hb.mi_cols + 255) / 256), dim3(256), 0, ctx->stream, dB);
    HIP_TRY(hipGetLastError());
    ED_STAGE(SVT_ENCDEC_STAGE_LF);
    svt_yuv_planes rec = p.recon;
    rec.width = width; rec.height = height;
    if (flags->apply_loop_filter) {
        hipLaunchKernelGGL(svt_lf_mask_kernel, dim3(hb.n_sb), dim3(64), 0, ctx->stream, dB, w->d_status);
        HIP_TRY(hipGetLastError());
        const svt_lf_mask *lfm = p.d_lfm;
        const int32_t      lfm_stride = hb.sb_cols, mr = hb.mi_rows, mc = hb.mi_cols;
        rc = svt_hip_lf_batch_device(ctx, 1, &rec, &lfm, &lfm_stride, thr, &mr, &mc, 0);
        if (rc) return rc;
    }
    ED_STAGE(SVT_ENCDEC_STAGE_PAD);
    if (flags->pad_reference && !p.no_pad) {
        rc = svt_hip_ref_pad_batch_device(ctx, 1, &rec, pad_x, pad_y);
        if (rc) return rc;
    }
    ED_STAGE(SVT_ENCDEC_STAGE_END);
    return SVT_HIP_OK;
}

extern "C" int32_t svt_hip_md_intra_default_device(svt_hip_ctx *ctx, int32_t width, int32_t height, int32_t filter_level, svt_lf_mode_info *d_lf_mi, int32_t mi_stride) {
    if (!ctx || !d_lf_mi || width < 8 || height < 8 || (width & 7) || (height & 7) || mi_stride < (width >> 3))
        return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "md_intra_default: bad argument");
    HIP_TRY(hipSetDevice(ctx->device));
    return svt_md_intra_default_launch(ctx, d_lf_mi, mi_stride, height >> 3, width >> 3, filter_level);
}

extern "C" void svt_hip_encdec_work_set_stage_hook(svt_encdec_work *w, svt_encdec_stage_hook hook, void *user) {
    if (!w) return;
    w->hook = hook;
    w->hook_user = user;
}

extern "C" int32_t svt_hip_encdec_work_status(svt_hip_ctx *ctx, svt_encdec_work *w, int32_t counts[8]) {
    if (!ctx || !w) return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "encdec_status: null");
    HIP_TRY(hipSetDevice(ctx->device));
    int32_t h[9];
    HIP_TRY(hipMemcpyAsync(h, w->d_off_cnt, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (counts) memcpy(counts, h, 8 * sizeof(int32_t));
    if (h[8]) {
        HIP_TRY(hipMemsetAsync(w->d_status, 0, sizeof(int32_t), ctx->stream));
        return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "encdec: malformed mode-info grid (block outside the picture, misaligned, or transform larger than its block)");
    }
    return SVT_HIP_OK;
}

extern "C" int32_t svt_hip_encdec_work_download(svt_hip_ctx *ctx, svt_encdec_work *w, svt_tq_block *blocks, uint32_t *pos, uint16_t *eob, int32_t capacity) {
    int32_t c[8];
    const int32_t rc = svt_hip_encdec_work_status(ctx, w, c);
    if (rc) return rc;
    const int32_t total = c[3] + c[7];
    if (total > capacity) return svt_set_error(SVT_HIP_ERR_NO_RESOURCES, "encdec_download: capacity");
    if (total > 0) {
        if (blocks) {
            if (!w->last_hb) return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "encdec_download: no batch yet");
            if (!w->d_blocks && hipMalloc((void **)&w->d_blocks, w->cap_per_pic * (size_t)w->max_pics * sizeof(svt_tq_block)) != hipSuccess) {
                w->d_blocks = nullptr;
                return svt_set_error(SVT_HIP_ERR_NO_RESOURCES, "encdec_download: device memory");
            }
            const ed_batch_dev *dB = nullptr;
            if (stage_batch(ctx, *(const ed_batch_dev *)w->last_hb, &dB)) return svt_set_error(SVT_HIP_ERR_NO_RESOURCES, "encdec_download: descriptor buffers");
            if (w->sb_order)
                hipLaunchKernelGGL(svt_tq_expand_sb_kernel, dim3(4 * w->last_pics * w->n_chunks), dim3(64), 0, ctx->stream, dB, (const int32_t *)w->d_counts, (const int32_t *)w->d_bases,
                                   (const uint32_t *)w->d_pos, w->d_blocks, w->n_chunks);
            else
            hipLaunchKernelGGL(svt_tq_expand_kernel, dim3((total + 255) / 256), dim3(256), 0, ctx->stream, dB, (const int32_t *)w->d_off_cnt, (const uint32_t *)w->d_pos, w->d_blocks);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpyAsync(blocks, w->d_blocks, (size_t)total * sizeof(svt_tq_block), hipMemcpyDeviceToHost, ctx->stream));
        }
        if (pos) HIP_TRY(hipMemcpyAsync(pos, w->d_pos, (size_t)total * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        if (eob) HIP_TRY(hipMemcpyAsync(eob, w->d_eob, (size_t)total * sizeof(uint16_t), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    return total;
}

extern "C" int32_t svt_hip_md_default_batch_device(svt_hip_ctx *ctx, int32_t n_pics, const svt_me_pu_result *const *d_results, int32_t width, int32_t height,
                                                   uint32_t lambda, int32_t filter_level, svt_mc_mode_info *const *d_mc_mi, svt_lf_mode_info *const *d_lf_mi,
                                                   int32_t mi_stride) {
    if (!ctx || n_pics < 1 || !d_results || !d_mc_mi || !d_lf_mi || width < 8 || height < 8 || (width & 7) || (height & 7) || mi_stride < (width >> 3))
        return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "md_default: bad argument");
    HIP_TRY(hipSetDevice(ctx->device));
    for (int first = 0; first < n_pics; first += ED_MAX_PICS) {
        const int n = n_pics - first < ED_MAX_PICS ? n_pics - first : ED_MAX_PICS;
        ed_batch_dev hb;
        memset(&hb, 0, sizeof hb);
        fill_dims(hb, n, width, height, mi_stride);
        hb.lambda = lambda; hb.filter_level = filter_level;
        for (int i = 0; i < n; i++) {
            if (!d_results[first + i] || !d_mc_mi[first + i] || !d_lf_mi[first + i]) return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "md_default: null picture");
            hb.pic[i].res = d_results[first + i]; hb.pic[i].mc_mi = d_mc_mi[first + i]; hb.pic[i].lf_mi = d_lf_mi[first + i];
        }
        const ed_batch_dev *dB = nullptr;
        if (stage_batch(ctx, hb, &dB)) return svt_set_error(SVT_HIP_ERR_NO_RESOURCES, "md_default: descriptor buffers");
        hipLaunchKernelGGL(svt_md_default_kernel, dim3(n * hb.n_sb), dim3(64), 0, ctx->stream, dB);
        HIP_TRY(hipGetLastError());
    }
    return SVT_HIP_OK;
}

extern "C" int32_t svt_hip_lf_build_masks_device(svt_hip_ctx *ctx, int32_t n_pics, const svt_lf_mode_info *const *d_lf_mi, int32_t mi_stride, int32_t mi_rows,
                                                 int32_t mi_cols, svt_lf_mask *const *d_lfm) {
    if (!ctx || n_pics < 1 || !d_lf_mi || !d_lfm || mi_rows < 1 || mi_cols < 1 || mi_stride < mi_cols) return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "lf_masks: bad argument");
    HIP_TRY(hipSetDevice(ctx->device));
    for (int first = 0; first < n_pics; first += ED_MAX_PICS) {
        const int n = n_pics - first < ED_MAX_PICS ? n_pics - first : ED_MAX_PICS;
        ed_batch_dev hb;
        memset(&hb, 0, sizeof hb);
        fill_dims(hb, n, mi_cols * 8, mi_rows * 8, mi_stride);
        for (int i = 0; i < n; i++) {
            if (!d_lf_mi[first + i] || !d_lfm[first + i]) return svt_set_error(SVT_HIP_ERR_BAD_PARAMETER, "lf_masks: null picture");
            hb.pic[i].lf_mi = (svt_lf_mode_info *)d_lf_mi[first + i]; hb.pic[i].lfm = d_lfm[first + i];
        }
        const ed_batch_dev *dB = nullptr;
        if (stage_batch(ctx, hb, &dB)) return svt_set_error(SVT_HIP_ERR_NO_RESOURCES, "lf_masks: descriptor buffers");
        hipLaunchKernelGGL(svt_lf_mask_kernel, dim3(n * hb.n_sb), dim3(64), 0, ctx->stream, dB, (int32_t *)nullptr);
        HIP_TRY(hipGetLastError());
    }
    return SVT_HIP_OK;
}
