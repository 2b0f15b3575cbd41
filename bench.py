#!/usr/bin/env python3
"""bench.py -- hot-path throughput of the MI355X implementation of SVT-VP9's block-level DSP path.

One "step" = one pass of the whole path over one mini-GOP (16 pictures of the 5-temporal-layer random-access structure
the reference uses at hierarchical_levels = 4) of synthetic 3840x2160 8-bit 4:2:0 input at the enc-mode 8 / tune 1 (OQ)
settings, q index 160 (-q 40).  Inside the timed region, per mini-GOP, every stage works on what the stage before it
wrote:

    picture analysis (padded + 1/16 planes from the source luma)  ->  motion estimation (16 B pictures, 2 lists)
    inter prediction from the mode-info grids  ->  residual / transform / quantisation / reconstruction + distortion
        + coefficient rate of the quantised blocks (one fused pass)  ->  in-loop deblocking of the reconstruction

The mode-info grids (partition, prediction direction and motion vectors of every block) are mode decision's output
-- host logic outside this path -- and are built ONCE before the timed loop from a first ME pass over the same
pictures (so the prediction the transform stage codes is the motion-compensated one), as are the loop-filter masks
(svt_hip_lf_build_masks on the same grids, skip flags from a first transform pass).  ME runs one mini-GOP ahead of the
EncDec-side stages, as the reference's ME threads do.  Inputs are resident in HBM before the timed region.
value = pictures / second (whole job, all ranks).

Contract: python bench.py --gpus N --steps K --warmup W ; for N>1 launched by torch.distributed.run, one rank per GPU;
GOP segments are independent (closed GOPs) so ranks share nothing and the only collectives are the
timing barrier / max-reduce.
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "tests"))

W4K, H4K = 3840, 2160
MINIGOP = 16
Q_INDEX = 160   # -q 40: quantizer_to_qindex[40]
# temporal layer of picture i (1..16) inside a 16-picture mini-GOP (5 layers, hierarchical_levels = 4)
LAYER = [4, 3, 4, 2, 4, 3, 4, 1, 4, 3, 4, 2, 4, 3, 4, 0]
STAGES = ("pa", "me", "mc", "tq", "rate", "lf")   # "rate" is a launch of its own only with SVT_BENCH_SEPARATE_RATE=1 (A/B aid)


def algorithmic_bytes_me(width, height, n_lists, l1_on):
    """src luma + src 1/16 (+1/4) + per list ref luma + ref 1/16 (+1/4) + results."""
    L = width * height
    nsb = ((width + 63) // 64) * ((height + 63) // 64)
    b = L * (1 + 1 / 16) + n_lists * L * (1 + 1 / 16) + 3400 * nsb
    if l1_on:
        b += (1 + n_lists) * L / 4
    return int(b)


def refs_of(i):
    """references inside the mini-GOP (display order): picture i at layer l is predicted from the nearest lower-layer
    pictures on both sides; the base-layer picture (16) from the previous base picture (0)"""
    if i == MINIGOP:
        return 0, 0
    span = MINIGOP >> LAYER[i - 1]
    return i - span, i + span


# -----------------------------------------------------------------------------------------------------------------------
# mode decision's output, synthesised: partition per 32x32 area, MVs / directions from the ME results
# -----------------------------------------------------------------------------------------------------------------------
def build_mode_info(B, res, kinds, mi_rows, mi_cols, nsbx):
    """res: ME results [n_sb][85] of the picture; kinds[area_row][area_col] in 0..3 = (8x8 blocks with 4x4 transforms, 8x8 / 8x8,
    16x16 / 16x16, 32x32 / 32x32).  Every block takes the best ME candidate of its own PU: direction (list 0 / list 1 /
    bi-prediction) and motion vectors (quarter-sample -> 1/8 sample)."""
    r, c = np.meshgrid(np.arange(mi_rows), np.arange(mi_cols), indexing="ij")
    k = kinds[r >> 2, c >> 2]
    q32 = ((r >> 2) & 1) * 2 + ((c >> 2) & 1)
    q16 = ((r >> 1) & 1) * 2 + ((c >> 1) & 1)
    q8 = (r & 1) * 2 + (c & 1)
    pu = np.where(k == 3, 1 + q32, np.where(k == 2, 5 + 4 * q32 + q16, 21 + 16 * q32 + 4 * q16 + q8))
    rec = res[(r >> 3) * nsbx + (c >> 3), pu]
    d = rec["dir0"].astype(np.int64)
    mi = np.zeros((mi_rows, mi_cols), dtype=B.MC_MODE_INFO_DTYPE)
    bw = np.where(k == 3, 4, np.where(k == 2, 2, 1)).astype(np.uint8)
    mi["bw8"], mi["bh8"] = bw, bw
    mi["ref_list"][..., 0] = np.where(d == 1, 1, 0)
    mi["ref_list"][..., 1] = np.where(d == 2, 1, -1)
    first_l1 = d == 1
    mi["mv_row"][..., 0] = 2 * np.where(first_l1, rec["y_mv_l1"], rec["y_mv_l0"])
    mi["mv_col"][..., 0] = 2 * np.where(first_l1, rec["x_mv_l1"], rec["x_mv_l0"])
    mi["mv_row"][..., 1] = np.where(d == 2, 2 * rec["y_mv_l1"].astype(np.int32), 0)
    mi["mv_col"][..., 1] = np.where(d == 2, 2 * rec["x_mv_l1"].astype(np.int32), 0)
    return mi, k


def build_tq_blocks(B, kinds, width, height, plane_w, iscan_off_dct):
    """transform blocks of one picture from its partition: luma transform = block size (4x4 inside 8x8 blocks of kind 0),
    chroma transform = uv_txsize_lookup of it (16x16 / 8x8 / 4x4 / 4x4); every sample of the 4:2:0 picture is covered by
    exactly one block.  Planes: Y rows, then U | V side by side (half width each) at the same stride."""
    out = [[] for _ in range(4)]   # per tx size: (row, col, plane) arrays

    def grid(r0, c0, n_area, n, limit_r):
        k = n_area // n
        rr = (r0[:, None, None] + (np.arange(k) * n)[None, :, None]).repeat(k, 2).ravel()
        cc = (c0[:, None, None] + (np.arange(k) * n)[None, None, :]).repeat(k, 1).ravel()
        keep = rr < limit_r
        return rr[keep], cc[keep]

    for kind in range(4):
        ay, ax = np.nonzero(kinds == kind)
        if not len(ay):
            continue
        n = (4, 8, 16, 32)[kind]
        rr, cc = grid(ay * 32, ax * 32, 32, n, height)
        out[kind].append((rr, cc))
        nuv, ts_uv = ((4, 0), (4, 0), (8, 1), (16, 2))[kind]
        for col0 in (0, width // 2):
            rr, cc = grid(ay * 16, ax * 16, 16, nuv, height // 2)
            out[ts_uv].append((height + rr, col0 + cc))
    arrs = []
    for ts in range(4):
        rr = np.concatenate([a for a, _ in out[ts]]) if out[ts] else np.zeros(0, np.int64)
        cc = np.concatenate([b for _, b in out[ts]]) if out[ts] else np.zeros(0, np.int64)
        a = np.zeros(len(rr), dtype=B.TQ_BLOCK_DTYPE)
        off = (rr * plane_w + cc).astype(np.uint32)
        a["src_off"] = a["pred_off"] = a["recon_off"] = off
        a["iscan_off"] = iscan_off_dct[ts]
        a["src_stride"] = a["pred_stride"] = a["recon_stride"] = plane_w
        a["tx_size"], a["tx_type"], a["do_recon"] = ts, 0, 1
        a["qtab"] = (rr >= height).astype(np.uint8)   # 0 luma, 1 chroma
        arrs.append(a)
    return arrs


def build_lf_mode_info(B, k_cell, nz4, mi_rows, mi_cols, level):
    """loop-filter view of the same partition: skip = the block has no non-zero luma coefficient (nz4: per 4x4 unit)"""
    pr, pc = (mi_rows + 3) // 4 * 4, (mi_cols + 3) // 4 * 4
    cell = np.zeros((pr, pc), bool)
    cell[:mi_rows, :mi_cols] = nz4[:2 * mi_rows, :2 * mi_cols].reshape(mi_rows, 2, mi_cols, 2).any(axis=(1, 3))
    b16 = np.kron(cell.reshape(pr // 2, 2, pc // 2, 2).any(axis=(1, 3)), np.ones((2, 2), bool))
    b32 = np.kron(cell.reshape(pr // 4, 4, pc // 4, 4).any(axis=(1, 3)), np.ones((4, 4), bool))
    nz = np.where(k_cell == 3, b32[:mi_rows, :mi_cols], np.where(k_cell == 2, b16[:mi_rows, :mi_cols], cell[:mi_rows, :mi_cols]))
    lmi = np.zeros((mi_rows, mi_cols), dtype=B.LF_MODE_INFO_DTYPE)
    lmi["sb_type"] = np.where(k_cell == 3, 9, np.where(k_cell == 2, 6, 3))
    lmi["tx_size"], lmi["skip"], lmi["is_inter"], lmi["filter_level"] = k_cell, ~nz, 1, level
    return lmi


# -----------------------------------------------------------------------------------------------------------------------
# CPU baseline: the oracle (kind "port"), built -O3 -march=native on this host, threaded over independent units
# -----------------------------------------------------------------------------------------------------------------------
def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def build_native_oracle():
    """oracle/*.c compiled for this host (auto-vectorised SAD / transform loops: the "AVX2-class" CPU proxy of SURVEY 8(d))"""
    out = os.path.join(tempfile.gettempdir(), f"liboracle_native_{os.getuid()}.so")
    src = [os.path.join(ROOT, "oracle", f) for f in ("oracle_me.c", "oracle_tq.c", "oracle_lf.c", "oracle_pa.c", "oracle_mc.c", "oracle_rate.c")]
    subprocess.check_call(["gcc", "-std=gnu11", "-O3", "-march=native", "-fPIC", "-shared", "-Wno-unused-function", "-o", out] + src + ["-lpthread"])
    return C.CDLL(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--width", type=int, default=W4K)
    ap.add_argument("--height", type=int, default=H4K)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--handoff", action="store_true", help="N > 1 only: split-GOP mode -- every step each rank also hands its padded base-layer reconstruction to "
                    "the next rank (RCCL send / recv over xGMI), the one exchange step of the path; closed GOPs (the default) need none")
    ap.add_argument("--stages", default=",".join(STAGES), help="profiling aid: run only these stages (the contract run uses all six)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    # one rank per GPU; if the launcher narrowed the visible devices to one per rank, LOCAL_RANK still counts from 0 upwards
    local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl")

    import importlib.util
    import me_configs as MC
    import svt_testlib as T
    _sp = importlib.util.spec_from_file_location("gop_shard", os.path.join(ROOT, "svt-vp9_amd", "gop_shard.py"))
    GS = importlib.util.module_from_spec(_sp)
    _sp.loader.exec_module(GS)
    B = T.B
    lib = B.load()
    dev = torch.device("cuda", local_rank)

    # streams: 0 = PA + ME (+ a second ME stream: ME of different pictures is independent, the tail of one launch is filled
    # by the other), 1 = EncDec side (inter prediction -> transform -> rate), 2 = deblocking.  ME is the long pole of the
    # step and fills every CU by itself (5 workgroups use all of a CU's LDS and 480 of the 512 registers per SIMD): the EncDec
    # stream gets the higher priority, so that when an ME workgroup retires a waiting transform workgroup moves in first
    # (measured: -1,0,0 = 3.70 ms/step, 0,0,0 = 3.64, 0,-1,0 = 3.61).
    prio = [int(x) for x in os.environ.get("SVT_BENCH_PRIO", "0,-1,0").split(",")]
    n_me_streams = max(1, int(os.environ.get("SVT_BENCH_ME_STREAMS", "2")))
    streams = [torch.cuda.Stream(device=local_rank, priority=prio[min(i, 2)]) for i in range(3)]
    streams += [torch.cuda.Stream(device=local_rank, priority=prio[0]) for _ in range(n_me_streams - 1)]
    ctxs = []
    for st_ in streams:
        c_ = C.c_void_p()
        B.check(lib.svt_hip_ctx_create_on_stream(C.byref(c_), local_rank, C.c_void_p(st_.cuda_stream)))
        ctxs.append(c_)
    ctx_me, ctx_enc, ctx_lf = ctxs[:3]
    me_ctxs, me_streams = [ctx_me] + ctxs[3:], [streams[0]] + streams[3:]

    Wd, Hd = args.width, args.height
    nsbx = (Wd + 63) // 64
    nsb = T.n_sb(Wd, Hd)
    mi_rows, mi_cols = Hd // 8, Wd // 8
    keep = []  # keeps device tensors alive

    def to_dev(a):
        t = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        keep.append(t)
        return t

    def dev_zeros(shape, dtype):
        t = torch.zeros(shape, dtype=dtype, device=dev)
        keep.append(t)
        return t

    # ---- synthetic mini-GOP (+ the previous base-layer picture), resident in HBM: Y rows followed by U | V rows (half
    # width each, same stride) in one buffer per picture ----
    frames = T.gen_clip(Wd, Hd, MINIGOP + 1, seed=GS.gop_seed(11, rank))  # rank r encodes its own GOP segment(s)
    plane_w, yuv_rows = Wd, Hd + Hd // 2
    pic_bytes = yuv_rows * plane_w
    src_all = np.zeros((MINIGOP + 1, yuv_rows, plane_w), np.uint8)   # index 0 = previous base picture
    for i, y in enumerate(frames):
        src_all[i, :Hd] = y
        src_all[i, Hd:, :Wd // 2] = y[::2, ::2] // 2 + 32
        src_all[i, Hd:, Wd // 2:] = 255 - y[::2, ::2] // 2 - y[1::2, 1::2] // 4
    d_src = to_dev(src_all)

    # ---- stage "pa": the three ME planes of every picture from its luma (device buffers laid out as the reference's
    # EbPaReferenceObject planes: padding 68 / 32 / 16) ----
    p_probe = B.me_params_preset(Wd, Hd, 8, 1, 2, 1, 4)
    l1_on = bool(p_probe.enable_hme_level_1_flag)
    pads = (68, 32, 16)

    def pa_alloc():
        d = B.PaPicture()
        for name, pad, sh in zip(("full", "quarter", "sixteenth"), pads, (0, 1, 2)):
            w, h = Wd >> sh, Hd >> sh
            t = dev_zeros((h + 2 * pad, w + 2 * pad), torch.uint8)
            pl = B.Plane()
            pl.buf, pl.stride, pl.origin_x, pl.origin_y, pl.width, pl.height = t.data_ptr(), w + 2 * pad, pad, pad, w, h
            setattr(d, name, pl)
        return d

    # two sets of analysed planes: picture analysis runs one mini-GOP ahead of motion estimation on its own stream (as the
    # reference's picture-analysis threads run ahead of its ME threads), writing set (k + 1) & 1 while ME reads set k & 1
    pa_sets = [[pa_alloc() for _ in range(MINIGOP + 1)] for _ in range(2)]
    pa_stream = torch.cuda.Stream(device=local_rank, priority=prio[0])
    ctx_pa = C.c_void_p()
    B.check(lib.svt_hip_ctx_create_on_stream(C.byref(ctx_pa), local_rank, C.c_void_p(pa_stream.cuda_stream)))
    ctxs.append(ctx_pa)

    def pa_call(ctx_, idx, s=0):
        n = len(idx)
        lum = (C.c_void_p * n)(*[d_src.data_ptr() + i * pic_bytes for i in idx])
        strides = (C.c_int32 * n)(*[plane_w] * n)
        out = (B.PaPicture * n)(*[pa_sets[s][i] for i in idx])
        B.check(lib.svt_hip_pa_prepare_batch_device(ctx_, n, lum, strides, out, 1 if l1_on else 0))

    for s_ in range(2):
        pa_call(ctx_me, [0], s_)   # the previous mini-GOP's base picture: analysed when that mini-GOP was
    B.check(lib.svt_hip_ctx_synchronize(ctx_me))
    pa_idx = list(range(1, MINIGOP + 1))

    # ---- stage "me": one batched launch per temporal layer ----
    results = [dev_zeros((nsb, 85 * 10), torch.int32) for _ in range(MINIGOP + 1)]
    # One launch per temporal layer, spread over the ME streams.  SVT_BENCH_ME_ONE_LAUNCH=1: the whole mini-GOP in one launch
    # (the per-picture parameters -- list count, temporal layer, same_ref_poc -- travel with the picture descriptors either
    # way: svt_hip_me_batch_layers_device).  Measured: ME alone 1.485 ms in one launch against 1.505 in five, but the whole
    # step 3.21 against 3.16 ms -- five launches on two streams interleave better with the EncDec-side kernels.
    per_layer = not os.environ.get("SVT_BENCH_ME_ONE_LAUNCH")
    me_launch_sets = []
    for pa_pics in pa_sets:
        me_launches = []
        if per_layer:
            groups = [[i for i in range(1, MINIGOP + 1) if LAYER[i - 1] == layer] for layer in range(5)]
        else:   # the whole mini-GOP in one launch, biggest search areas (lowest layers) first
            groups = [sorted(range(1, MINIGOP + 1), key=lambda i: LAYER[i - 1])]
        for idx in groups:
            n = len(idx)
            p = (B.MeParams * n)()
            for k_, i in enumerate(idx):
                p[k_] = B.me_params_preset(Wd, Hd, 8, 1, 2, LAYER[i - 1], 4)
                p[k_].same_ref_poc = 1 if LAYER[i - 1] == 0 else 0
            cur = (B.PaPicture * n)(*[pa_pics[i] for i in idx])
            r0 = (B.PaPicture * n)(*[pa_pics[refs_of(i)[0]] for i in idx])
            r1 = (B.PaPicture * n)(*[pa_pics[refs_of(i)[1]] for i in idx])
            res = (C.c_void_p * n)(*[results[i].data_ptr() for i in idx])
            me_launches.append((n, cur, r0, r1, p, res))
        me_launch_sets.append(me_launches)
    # launches -> ME streams: largest first onto the least loaded stream
    me_slot, load = [0] * len(me_launches), [0] * len(me_ctxs)
    for li in sorted(range(len(me_launches)), key=lambda j: -me_launches[j][0]):
        k_ = load.index(min(load))
        me_slot[li] = k_
        load[k_] += me_launches[li][0]
    # per-launch durations of the dominant kernel: a (start, stop) HIP event pair around EVERY timed ME launch, recorded on the
    # stream the launch goes to.  The events come from a pool created before the timed region: creating an event costs the host
    # tens of microseconds, and ten creations per step in the launch path cost the overlap of the two ME streams (ME alone 4.46
    # instead of 2.26 ms per step, measured); recording an existing event costs nothing measurable.
    me_ev = []
    me_pool = [torch.cuda.Event(enable_timing=True) for _ in range(2 * len(me_launches) * (args.steps + 1))]
    for e_ in me_pool:
        e_.record(me_streams[0])   # an event object is created on its first record

    def run_me(record=False, s=0):
        for st_ in me_streams[1:]:
            st_.wait_stream(me_streams[0])
        for li, (n, cur, r0, r1, p, res) in enumerate(me_launch_sets[s]):
            k_ = me_slot[li]
            if record:
                e0, e1 = me_pool.pop(), me_pool.pop()
                e0.record(me_streams[k_])
            B.check(lib.svt_hip_me_batch_layers_device(me_ctxs[k_], n, cur, r0, r1, p, res, None))
            if record:
                e1.record(me_streams[k_])
                me_ev.append((e0, e1))
        for st_ in me_streams[1:]:
            me_streams[0].wait_stream(st_)

    # ---- first pass (setup, untimed): PA + ME, results to the host = the input of the synthesised mode decision ----
    pa_call(ctx_me, pa_idx)
    with torch.cuda.stream(streams[0]):
        run_me()
    for c_ in ctxs:
        B.check(lib.svt_hip_ctx_synchronize(c_))
    torch.cuda.synchronize()
    rng = np.random.default_rng(5 + rank)
    area_rows, area_cols = (Hd + 31) // 32, (Wd + 31) // 32
    iscan, ioffs = T.iscan_array()
    iscan_off_dct = [ioffs[(ts, 0)] for ts in range(4)]
    mi_list, kcell_list, blocks_by_ts = [], [], [[] for _ in range(4)]
    for i in range(1, MINIGOP + 1):
        res = results[i].cpu().numpy().view(B.ME_RESULT_DTYPE).reshape(nsb, 85)
        kinds = rng.integers(0, 4, (area_rows, area_cols))
        if Hd % 32:   # blocks must not reach below the picture
            kinds[-1] = np.minimum(kinds[-1], 2 if Hd % 32 == 16 else 1)
        mi, k_cell = build_mode_info(B, res, kinds, mi_rows, mi_cols, nsbx)
        mi_list.append(mi)
        kcell_list.append(k_cell)
        for ts, a in enumerate(build_tq_blocks(B, kinds, Wd, Hd, plane_w, iscan_off_dct)):
            for f in ("src_off", "pred_off", "recon_off"):
                a[f] += np.uint32((i - 1) * pic_bytes)
            a["src_off"] += np.uint32(pic_bytes)        # d_src also holds picture 0
            blocks_by_ts[ts].append(a)
    # the whole mini-GOP is one batch, grouped by transform size across pictures -> 4 launches, each fills the GPU
    tq_blocks_all = np.concatenate([a for ts in range(4) for a in blocks_by_ts[ts]])
    counts = [sum(len(a) for a in blocks_by_ts[ts]) for ts in range(4)]
    pic_of_block = np.concatenate([np.full(len(a), k, np.int32) for ts in range(4) for k, a in enumerate(blocks_by_ts[ts])])
    nn = (16 << (2 * tq_blocks_all["tx_size"].astype(np.int64)))
    tq_blocks_all["coeff_off"] = np.concatenate([[0], np.cumsum(nn)[:-1]]).astype(np.uint32)
    n_coeff_all = int(nn.sum())
    cnt_c = (C.c_int32 * 4)(*counts)

    # quantiser tables of q index 160, luma and chroma (no chroma deltas): the steps are the reference's eb_vp9_dc_quant /
    # eb_vp9_ac_quant values (committed fixture), the tables svt_hip_quant_tables_init's
    qrow = np.load(os.path.join(T.GOLDEN_DIR, "quant_reference.npz"))["0|0|0"][Q_INDEX]
    qtabs = np.zeros(2, dtype=B.QUANT_DTYPE)
    for j, base in enumerate((2, 14)):
        B.check(lib.svt_hip_quant_tables_init(Q_INDEX, int(qrow[1]), int(qrow[base]), int(qrow[base + 1]), qtabs[j:j + 1].ctypes.data_as(C.c_void_p)))
    ac_q = int(qrow[3])

    # ---- stage "mc": inter prediction of the 16 pictures from their mode-info grids; the references are the padded
    # pictures of the reference lists (80 / 40 samples of padding, Codec/EbEncHandle.c:968-971) ----
    pad = 80
    ref_pics = []
    for i in range(MINIGOP + 1):
        y = src_all[i, :Hd]
        u, v = src_all[i, Hd:, :Wd // 2], src_all[i, Hd:, Wd // 2:]
        ty, tu, tv = (to_dev(np.pad(pl, pd, mode="edge")) for pl, pd in ((y, pad), (u, pad // 2), (v, pad // 2)))
        ref_pics.append((ty.data_ptr() + pad * ty.shape[1] + pad, tu.data_ptr() + (pad // 2) * tu.shape[1] + pad // 2,
                         tv.data_ptr() + (pad // 2) * tv.shape[1] + pad // 2, ty.shape[1], tu.shape[1]))
    d_mi = [to_dev(m.view(np.uint8)) for m in mi_list]
    d_pred = dev_zeros((MINIGOP, yuv_rows, plane_w), torch.uint8)
    mc_pics = (B.McPicture * MINIGOP)()
    for k in range(MINIGOP):
        mp = mc_pics[k]
        mp.d_mi, mp.mi_stride, mp.mi_rows, mp.mi_cols, mp.use_subpel = d_mi[k].data_ptr(), mi_cols, mi_rows, mi_cols, 1
        for l in range(2):
            r = mp.ref[l]
            r.y, r.u, r.v, r.y_stride, r.uv_stride = ref_pics[refs_of(k + 1)[l]]
            r.width, r.height = Wd, Hd
        base = d_pred.data_ptr() + k * pic_bytes
        mp.pred.y, mp.pred.u, mp.pred.v = base, base + Hd * plane_w, base + Hd * plane_w + Wd // 2
        mp.pred.y_stride, mp.pred.uv_stride, mp.pred.width, mp.pred.height = plane_w, plane_w, Wd, Hd

    def run_mc():
        B.check(lib.svt_hip_inter_pred_batch_device(ctx_enc, MINIGOP, mc_pics))

    # ---- stage "tq": residual -> transform -> quantisation -> reconstruction (+ coefficient-domain distortion) ----
    rate_ctx = np.random.default_rng(8).integers(0, 3, len(tq_blocks_all)).astype(np.uint8)   # entropy context of every block (an input)
    tq_blocks_all["pad"][:, 0] = rate_ctx | (tq_blocks_all["qtab"] << 2) | (1 << 3)            # SVT_TQ_RATE_INFO(ctx, plane_type, is_inter = 1)
    d_blocks, d_qt, d_iscan = to_dev(tq_blocks_all.view(np.uint8)), to_dev(qtabs.view(np.uint8)), to_dev(iscan)
    d_q, d_dq = dev_zeros(n_coeff_all, torch.int16), dev_zeros(n_coeff_all, torch.int16)
    d_eob = dev_zeros(len(tq_blocks_all), torch.int16)
    d_dist = dev_zeros(2 * len(tq_blocks_all), torch.int64)
    d_rec = [dev_zeros((MINIGOP, yuv_rows, plane_w), torch.uint8) for _ in range(2)]   # double-buffered: LF(k) || TQ(k+1)
    # the block list is the same for both reconstruction buffers (offsets are relative to the buffer)

    separate_rate = os.environ.get("SVT_BENCH_SEPARATE_RATE", "0") == "1"

    def run_tq_plain(buf):
        B.check(lib.svt_hip_tq_batch_dist_device(ctx_enc, C.c_void_p(d_src.data_ptr()), C.c_void_p(d_pred.data_ptr()), C.c_void_p(d_rec[buf].data_ptr()),
                                                 C.c_void_p(d_blocks.data_ptr()), cnt_c, C.c_void_p(d_qt.data_ptr()), C.c_void_p(d_iscan.data_ptr()),
                                                 C.c_void_p(d_q.data_ptr()), C.c_void_p(d_dq.data_ptr()), C.c_void_p(d_eob.data_ptr()),
                                                 C.c_void_p(d_dist.data_ptr())))

    # first transform pass (setup): eobs -> skip flags of the loop-filter mode info and the rate stage's records
    with torch.cuda.stream(streams[1]):
        run_mc()
        run_tq_plain(0)
    B.check(lib.svt_hip_ctx_synchronize(ctx_enc))
    torch.cuda.synchronize()
    eob_h = d_eob.cpu().numpy().view(np.uint16)
    # what the synthesised mode decision produced (reported with the result: the transform / rate stages' work depends on it)
    with torch.no_grad():
        resid = (d_src[1:, :Hd].to(torch.int16) - d_pred[:, :Hd].to(torch.int16)).abs().to(torch.float32).mean().item()
    workload_stats = {"mean_abs_luma_residual": round(resid, 2),
                      "mean_eob_by_tx_size": [round(float(eob_h[tq_blocks_all["tx_size"] == ts].mean()), 1) for ts in range(4)],
                      "blocks_by_tx_size": counts}

    # ---- stage "rate": bits of every transform block from the quantised coefficients the transform stage wrote ----
    rtab, rscan = T.rate_tables()
    roffs, _ = T.rate_scan_offsets()
    rb = np.zeros(len(tq_blocks_all), dtype=B.RATE_BLOCK_DTYPE)
    rb["coeff_off"], rb["tx_size"], rb["eob"] = tq_blocks_all["coeff_off"], tq_blocks_all["tx_size"], eob_h
    rb["scan_off"] = np.array([roffs[(ts, 0)] for ts in range(4)], np.uint32)[tq_blocks_all["tx_size"]]
    rb["plane_type"], rb["is_inter"] = tq_blocks_all["qtab"], 1
    rb["ctx"] = rate_ctx
    d_rb, d_rt, d_rs = to_dev(rb.view(np.uint8)), to_dev(np.ascontiguousarray(rtab).reshape(1).view(np.uint8)), to_dev(rscan)
    d_bits = dev_zeros(len(rb), torch.int32)

    def run_tq_rd(buf):   # distortion + rate behind the quantiser: perform_dist_rate_calc in one pass
        B.check(lib.svt_hip_tq_rd_batch_device(ctx_enc, C.c_void_p(d_src.data_ptr()), C.c_void_p(d_pred.data_ptr()), C.c_void_p(d_rec[buf].data_ptr()),
                                               C.c_void_p(d_blocks.data_ptr()), cnt_c, C.c_void_p(d_qt.data_ptr()), C.c_void_p(d_iscan.data_ptr()),
                                               C.c_void_p(d_q.data_ptr()), C.c_void_p(d_dq.data_ptr()), C.c_void_p(d_eob.data_ptr()),
                                               C.c_void_p(d_dist.data_ptr()), C.c_void_p(d_rt.data_ptr()), C.c_void_p(d_rs.data_ptr()),
                                               C.c_void_p(d_bits.data_ptr())))

    run_tq = run_tq_plain if separate_rate else run_tq_rd

    def run_rate():
        B.check(lib.svt_hip_coeff_rate_batch_device(ctx_enc, C.c_void_p(d_q.data_ptr()), C.c_void_p(d_rb.data_ptr()), len(rb), C.c_void_p(d_rt.data_ptr()),
                                                    C.c_void_p(d_rs.data_ptr()), C.c_void_p(d_bits.data_ptr())))

    # ---- stage "lf": in-loop deblocking of the 16 reconstructed pictures, one batched launch; per-picture masks from the
    # pictures' own mode info (svt_hip_lf_build_masks), filter level from the q index ----
    level = lib.svt_hip_lf_level_from_q(ac_q, 0)
    thr = B.LfThresh()
    lib.svt_hip_lf_thresh_init(C.byref(thr), 0)
    sb_rows, sb_cols = (mi_rows + 7) // 8, (mi_cols + 7) // 8
    d_lfm = []
    luma = (tq_blocks_all["qtab"] == 0)
    for k in range(MINIGOP):
        sel = luma & (pic_of_block == k)
        off = tq_blocks_all["recon_off"][sel].astype(np.int64) - k * pic_bytes
        r4, c4 = (off // plane_w) >> 2, (off % plane_w) >> 2
        n4 = (1 << tq_blocks_all["tx_size"][sel].astype(np.int64))
        nz4 = np.zeros(((Hd + 31) // 32 * 8, (Wd + 31) // 32 * 8), bool)
        nzb = eob_h[sel] != 0
        for s in range(4):   # a block of 2^s x 2^s 4x4 units
            m = nzb & (n4 == (1 << s))
            for dy in range(1 << s):
                for dx in range(1 << s):
                    nz4[r4[m] + dy, c4[m] + dx] = True
        lmi = build_lf_mode_info(B, kcell_list[k], nz4, mi_rows, mi_cols, level)
        lfm = np.zeros((sb_rows, sb_cols), dtype=B.LF_MASK_DTYPE)
        B.check(lib.svt_hip_lf_build_masks(lmi.ctypes.data_as(C.c_void_p), mi_cols, mi_rows, mi_cols, lfm.ctypes.data_as(C.c_void_p), sb_cols))
        d_lfm.append(to_dev(lfm.view(np.uint8)))
    lfm_ptrs = (C.c_void_p * MINIGOP)(*[t.data_ptr() for t in d_lfm])
    i32 = lambda v: (C.c_int32 * MINIGOP)(*[v] * MINIGOP)
    lfs, mrs, mcs = i32(sb_cols), i32(mi_rows), i32(mi_cols)
    lf_desc = []
    for buf in range(2):
        dsc = (B.YuvPlanes * MINIGOP)()
        for k in range(MINIGOP):
            base = d_rec[buf].data_ptr() + k * pic_bytes
            d = dsc[k]
            d.y, d.u, d.v = base, base + Hd * plane_w, base + Hd * plane_w + Wd // 2
            d.y_stride, d.uv_stride, d.width, d.height = plane_w, plane_w, Wd, Hd
        lf_desc.append(dsc)

    def run_lf(buf):
        B.check(lib.svt_hip_lf_batch_device(ctx_lf, MINIGOP, lf_desc[buf], lfm_ptrs, lfs, C.byref(thr), mrs, mcs, 0))

    # split-GOP hand-off (optional, N > 1): the padded base-layer reconstruction (luma + chroma with 80 / 40 samples of padding)
    handoff = args.handoff and world > 1
    if handoff:
        ho_bytes = (Wd + 2 * pad) * (Hd + 2 * pad) + 2 * (Wd // 2 + pad) * (Hd // 2 + pad)
        ho_send, ho_recv = dev_zeros(ho_bytes, torch.uint8), dev_zeros(ho_bytes, torch.uint8)
        ho_stream = torch.cuda.Stream(device=local_rank)

    def run_handoff():
        # ordered after this step's deblocking (the picture handed over is its output), overlapped with the next step's work
        ho_stream.wait_stream(streams[2])
        with torch.cuda.stream(ho_stream):
            ops = [dist.P2POp(dist.isend, ho_send, (rank + 1) % world), dist.P2POp(dist.irecv, ho_recv, (rank - 1) % world)]
            for w_ in dist.batch_isend_irecv(ops):
                w_.wait()

    stages = set(args.stages.split(","))
    ev = []            # (stage name, start event, stop event) of every stage of every timed step
    lf_done = [None, None]   # event after the deblocking that last read reconstruction buffer b
    pa_done, me_done = [None, None], [None, None]   # analysed-plane set s: written / last read

    def staged(name, stream, fn, record):
        if name not in stages:
            return
        if not record:
            return fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        fn()
        e1.record(stream)
        ev.append((name, e0, e1))

    step_no = [0]

    def step(record=False):
        buf = step_no[0] & 1
        step_no[0] += 1
        # ME side (one mini-GOP ahead of the EncDec side, as the reference's ME threads are): PA writes the planes ME reads
        # and picture analysis of the NEXT mini-GOP runs beside it on its own stream
        if pa_done[buf] is not None:
            streams[0].wait_event(pa_done[buf])
        staged("me", streams[0], lambda: run_me(record, buf), record)
        if "me" in stages:
            me_done[buf] = torch.cuda.Event()
            me_done[buf].record(streams[0])
        if "pa" in stages:
            if me_done[1 - buf] is not None:
                pa_stream.wait_event(me_done[1 - buf])   # ME of the previous step read the set written now
            staged("pa", pa_stream, lambda: pa_call(ctx_pa, pa_idx, 1 - buf), record)
            pa_done[1 - buf] = torch.cuda.Event()
            pa_done[1 - buf].record(pa_stream)
        # EncDec side: prediction -> transform (reads the prediction, writes coefficients + reconstruction) -> rate (reads the
        # coefficients); the reconstruction buffer is free again once the deblocking that used it two steps ago is done
        if lf_done[buf] is not None:
            streams[1].wait_event(lf_done[buf])
        staged("mc", streams[1], run_mc, record)
        staged("tq", streams[1], lambda: run_tq(buf), record)
        e_tq = torch.cuda.Event()
        e_tq.record(streams[1])
        if separate_rate:
            staged("rate", streams[1], run_rate, record)
        streams[2].wait_event(e_tq)
        staged("lf", streams[2], lambda: run_lf(buf), record)
        lf_done[buf] = torch.cuda.Event()
        lf_done[buf].record(streams[2])
        if handoff:
            run_handoff()

    def sync():
        for c_ in ctxs:
            B.check(lib.svt_hip_ctx_synchronize(c_))
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    # host cost of a step = time to enqueue it while nothing holds the enqueue thread back: over the first steps only -- once the
    # 64-slot descriptor rings of the contexts are full (~20 steps ahead of the GPU) the thread waits for the GPU and the
    # figure would read as the GPU's step time
    n_free = min(args.steps, 16)
    for i_ in range(args.steps):
        step(record=True)
        if i_ + 1 == n_free:
            t_enqueued = time.perf_counter() - t0
    sync()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    dt = GS.reduce_elapsed(dt, dist if world > 1 else None, dev)

    # ---- per-stage time: HIP events on each stage's own stream, bracketing that stage's launches of every timed step (so
    # it includes whatever slowdown the overlap with the other stages causes, like a rocprofv3 trace) ----
    stage_ms = {s: 0.0 for s in STAGES}
    for name, e0, e1 in ev:
        stage_ms[name] += e0.elapsed_time(e1) / args.steps
    me_launch_ms = sum(e0.elapsed_time(e1) for e0, e1 in me_ev)   # sum of the individual launch durations (the two ME streams overlap)
    n_me_launch = max(1, len(me_ev))
    L = Wd * Hd
    inter = np.concatenate([(m["ref_list"][..., 0] >= 0).ravel() for m in mi_list])
    comp = np.concatenate([(m["ref_list"][..., 1] >= 0).ravel() for m in mi_list])
    stage_bytes = {
        # source luma read + padded / decimated planes written
        "pa": MINIGOP * int(L + (Wd + 2 * pads[0]) * (Hd + 2 * pads[0]) + (Wd // 4 + 2 * pads[2]) * (Hd // 4 + 2 * pads[2]) +
                            (((Wd // 2 + 2 * pads[1]) * (Hd // 2 + 2 * pads[1])) if l1_on else 0)),
        "me": MINIGOP * algorithmic_bytes_me(Wd, Hd, 2, l1_on),
        # per 8x8 unit: 96 bytes (64 luma + 2 x 16 chroma) read per reference and written once, + the 12-byte mode-info record
        "mc": int(96 * (inter.sum() + (inter & comp).sum()) + 96 * inter.sum() + 12 * inter.size),
        "tq": MINIGOP * int(7.5 * L),                # src 1.5L + pred 1.5L + qcoeff 3L + recon 1.5L (the kernel also writes dqcoeff, +3L)
        # per block: the coefficients up to eob (whole 64-byte lines), 16-byte descriptor, 4-byte result
        "rate": int(np.sum(np.minimum(nn, ((eob_h.astype(np.int64) * 2 + 63) // 64 + 1) * 32)) * 2 + 20 * len(rb)),
        "lf": MINIGOP * (3 * L + 160 * nsb),         # recon read + write (3L) + masks
    }
    kernel_of = {"pa": "svt_pa_plane_kernel", "me": "svt_me_sb_kernel", "mc": "svt_mc_kernel",
                 "tq": "svt_tq_kernel<4|8|16|32>" + ("" if separate_rate else " (+ fused coefficient rate)"),
                 "rate": "svt_rate_kernel", "lf": "svt_lf_kernel"}
    if not separate_rate:
        stages.discard("rate")
    if rank != 0:
        return
    me_ms = max(stage_ms["me"], 1e-9)
    # roofline of the dominant kernel, per launch: algorithmic bytes of a launch / its own duration, averaged over the
    # launches = (bytes of all launches) / (sum of their durations); the stream-span figure is given beside it
    per_launch_ms = me_launch_ms / n_me_launch
    achieved = stage_bytes["me"] * args.steps / max(me_launch_ms * 1e-3, 1e-12) / 1e9 if me_ev else 0.0
    traffic, valu = None, None
    tj = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tj) and (Wd, Hd) == (W4K, H4K):
        rec = json.load(open(tj)).get("svt_me_sb_kernel", {})
        if rec.get("bytes_per_step"):
            traffic = int(rec["bytes_per_step"] / len(me_launches))   # per launch, like `achieved`
        vi = rec.get("valu_wave_insts_per_step")
        if vi and "me" in stages:
            # the kernel's real roof: 64-lane VALU instructions issued (rocprofv3 SQ_INSTS_VALU, profiles/) per second against
            # 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz (MI355X_MICROARCH.md)
            ach = vi * 64 / (me_ms * 1e-3) / 1e12
            valu = {"achieved": round(ach, 2), "peak": 39.3, "unit": "T lane-ops/s", "frac": round(ach / 39.3, 4), "wave_insts_per_step": vi}
    fps = GS.aggregate_rate(MINIGOP, args.steps, world, dt)
    out = {
        "metric": "encoded frames/sec (block-level DSP hot path: picture analysis + ME + inter prediction + DCT/quant/recon + "
                  "coefficient rate + deblock), 4Kp60 yuv420p enc-mode 8",
        "value": round(fps, 2),
        "unit": "frames/s",
        "mpixels_per_s": round(fps * Wd * Hd / 1e6, 1),
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3),
        "host_enqueue_ms_per_step": round(t_enqueued / n_free * 1e3, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u8",
        "data": "synthetic",
        "config": {"workload": f"{Wd}x{Hd} 8-bit yuv420p, -enc-mode 8 -tune 1 -q 40, 1xMI355X per rank; step = one 16-picture "
                               "mini-GOP (5 temporal layers, B pictures, 2 reference lists) through the hot path, every stage "
                               "consuming the previous stage's output",
                   "stages": ["picture_analysis", "motion_estimation", "inter_prediction", "transform_quant_recon_distortion",
                              "coefficient_rate", "deblocking"],
                   "stages_run": [s for s in STAGES if s in stages],
                   "pictures_per_step": MINIGOP, "transform_blocks_per_step": int(len(tq_blocks_all)), "q_index": Q_INDEX,
                   "workload_stats": workload_stats,
                   "parallelism": f"gop-shard x{world}" + (" + split-GOP reference hand-off (RCCL send/recv)" if handoff else "")},
        "roofline": {"bound": "hbm", "kernel": "svt_me_sb_kernel", "achieved": round(achieved, 2), "peak": 8000.0,
                     "unit": "GB/s", "frac": round(achieved / 8000.0, 5), "traffic": traffic,
                     "launches_per_step": len(me_launches), "avg_launch_ms": round(per_launch_ms, 4),
                     "algorithmic_bytes_per_launch": int(stage_bytes["me"] / len(me_launches)),
                     "stream_span": {"kernel_ms_per_step": round(me_ms, 3), "GB_per_s": round(stage_bytes["me"] / (me_ms * 1e-3) / 1e9, 2),
                                     "frac": round(stage_bytes["me"] / (me_ms * 1e-3) / 8e12, 5),
                                     "note": f"{n_me_streams} ME streams run launches concurrently: each launch's own duration is stretched"},
                     "valu": valu},
        "kernels": {kernel_of[s]: {"ms_per_step": round(stage_ms[s], 3), "algorithmic_bytes_per_step": stage_bytes[s],
                                   "GB_per_s": round(stage_bytes[s] / (max(stage_ms[s], 1e-9) * 1e-3) / 1e9, 2),
                                   "frac_of_8TBps": round(stage_bytes[s] / (max(stage_ms[s], 1e-9) * 1e-3) / 8e12, 5)}
                    for s in STAGES if s in stages},
    }
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(T, B, frames, src_all, mi_list, tq_blocks_all, pic_of_block, qtabs, iscan, rb, rtab, rscan,
                                           [np.frombuffer(t.cpu().numpy(), dtype=B.LF_MASK_DTYPE).reshape(sb_rows, sb_cols) for t in d_lfm],
                                           thr, Wd, Hd, plane_w, l1_on, n_coeff_all)
    print(json.dumps(out))
    for c_ in ctxs:
        lib.svt_hip_ctx_destroy(c_)


def cpu_baseline(T, B, frames, src_all, mi_list, tq_blocks_all, pic_of_block, qtabs, iscan, rb, rtab, rscan, lfms, thr, Wd, Hd,
                 plane_w, l1_on, n_coeff_all):
    """The oracle (C restatement of the reference's C path; kind "port"), compiled -O3 -march=native on this host and spread
    over host threads, on the SAME workload: a bounded sample of whole pictures of the mini-GOP through all six stages.
    Timed with all host cores and with 8.  Reported, never the target."""
    from concurrent.futures import ThreadPoolExecutor
    orc = build_native_oracle()
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    ncpu = os.cpu_count() or 1
    sample = [8, 4, 2, 1, 3, 12, 5, 7]   # mini-GOP positions: temporal layers 1, 2, 3, 4, 4, 2, 4, 4 (half of a mini-GOP is layer 4)
    mi_rows, mi_cols = Hd // 8, Wd // 8
    nsb = T.n_sb(Wd, Hd)
    yuv_rows = Hd + Hd // 2
    pic_bytes = yuv_rows * plane_w
    pad = 80

    def run(nthreads):
        t = {}
        with ThreadPoolExecutor(nthreads) as ex:
            # picture analysis: one picture per task (the three padded planes of every picture ME touches)
            need = sorted({j for i in sample for j in (i,) + refs_of(i)})

            class HostPa:
                def __init__(self, luma):
                    self.arr = [np.zeros(((Hd >> sh) + 2 * pd, (Wd >> sh) + 2 * pd), np.uint8) for sh, pd in ((0, 68), (1, 32), (2, 16))]
                    self.d = B.PaPicture()
                    self.d.full, self.d.quarter, self.d.sixteenth = (B.plane_desc(a, pd, pd) for a, pd in zip(self.arr, (68, 32, 16)))
                    self.luma = np.ascontiguousarray(luma)

                def desc(self):
                    return self.d
            pics = {j: HostPa(frames[j]) for j in need}

            def pa_one(j):
                rc = orc.svt_oracle_pa_prepare(vp(pics[j].luma), Wd, C.byref(pics[j].d), 1 if l1_on else 0)
                assert rc == 0
            t0 = time.perf_counter()
            list(ex.map(pa_one, need))
            t["pa"] = (time.perf_counter() - t0) * len(sample) / len(need)
            # motion estimation: SB ranges of the sampled pictures
            jobs = []
            res = {i: np.zeros((nsb, 85), dtype=B.ME_RESULT_DTYPE) for i in sample}
            step = max(4, -(-len(sample) * nsb // (3 * nthreads)))   # ~3 jobs per thread: the oracle allocates its planes per call
            for i in sample:
                p = B.me_params_preset(Wd, Hd, 8, 1, 2, LAYER[i - 1], 4)
                a, b = refs_of(i)
                dc, d0, d1 = pics[i].desc(), pics[a].desc(), pics[b].desc()
                for s0 in range(0, nsb, step):
                    jobs.append((dc, d0, d1, p, res[i], s0, min(nsb, s0 + step)))
            t0 = time.perf_counter()
            list(ex.map(lambda j: orc.svt_oracle_me_picture(C.byref(j[0]), C.byref(j[1]), C.byref(j[2]), C.byref(j[3]), vp(j[4]), None, j[5], j[6]), jobs))
            t["me"] = time.perf_counter() - t0
            # inter prediction: one picture per task (the oracle walks the mode-info grid)
            refs_h = {}
            for j in sorted({j for i in sample for j in refs_of(i)}):
                y, u, v = src_all[j, :Hd], src_all[j, Hd:, :Wd // 2], src_all[j, Hd:, Wd // 2:]
                refs_h[j] = tuple(np.ascontiguousarray(np.pad(pl, pd, mode="edge")) for pl, pd in ((y, pad), (u, pad // 2), (v, pad // 2)))
            preds = {}

            def mc_one(i):
                hr = (B.McHostRef * 2)()
                for l, j in enumerate(refs_of(i)):
                    y, u, v = refs_h[j]
                    hr[l].y, hr[l].u, hr[l].v, hr[l].y_stride, hr[l].uv_stride, hr[l].org_x, hr[l].org_y = (y.ctypes.data, u.ctypes.data, v.ctypes.data,
                                                                                                        y.shape[1], u.shape[1], pad, pad)
                out = [np.zeros((Hd, Wd), np.uint8), np.zeros((Hd // 2, Wd // 2), np.uint8), np.zeros((Hd // 2, Wd // 2), np.uint8)]
                mi = np.ascontiguousarray(mi_list[i - 1])
                rc = orc.svt_oracle_inter_pred_frame(vp(mi), mi_cols, mi_rows, mi_cols, hr, 1, *[vp(o) for o in out])
                assert rc == 0
                preds[i] = out
            t0 = time.perf_counter()
            list(ex.map(mc_one, sample))
            t["mc"] = time.perf_counter() - t0
            # transform / quantisation / reconstruction: block ranges of the sampled pictures
            q_h, dq_h = np.zeros(n_coeff_all, np.int16), np.zeros(n_coeff_all, np.int16)
            eob_o = np.zeros(len(tq_blocks_all), np.uint16)
            pred_h = np.zeros((MINIGOP, yuv_rows, plane_w), np.uint8)
            rec_h = np.zeros((MINIGOP, yuv_rows, plane_w), np.uint8)
            for i in sample:
                y, u, v = preds[i]
                pred_h[i - 1, :Hd], pred_h[i - 1, Hd:, :Wd // 2], pred_h[i - 1, Hd:, Wd // 2:] = y, u, v
            blk = tq_blocks_all.copy()
            blk["src_off"] -= np.uint32(pic_bytes)       # host source buffer below starts at picture 1
            src_h = src_all[1:]
            sel = np.nonzero(np.isin(pic_of_block, [i - 1 for i in sample]))[0]
            chunks = [ix for ix in np.array_split(sel, max(1, 3 * nthreads)) if len(ix)]
            blk_of = [np.ascontiguousarray(blk[ix]) for ix in chunks]     # sliced outside the timed part
            rb_of = [np.ascontiguousarray(rb[ix]) for ix in chunks]

            def tq_chunk(k):
                ix, b = chunks[k], blk_of[k]
                e = np.zeros(len(ix), np.uint16)
                rc = orc.svt_oracle_tq_batch(vp(src_h), vp(pred_h), vp(rec_h), vp(b), len(ix), vp(qtabs), vp(iscan), vp(q_h), vp(dq_h), vp(e))
                assert rc == 0
                eob_o[ix] = e
            t0 = time.perf_counter()
            list(ex.map(tq_chunk, range(len(chunks))))
            t["tq"] = time.perf_counter() - t0
            # coefficient rate: block ranges
            bits = np.zeros(len(rb), np.int32)

            rtab_c = np.ascontiguousarray(rtab).reshape(1)

            def rate_chunk(k):
                ix, r = chunks[k], rb_of[k]
                r["eob"] = eob_o[ix]
                o = np.zeros(len(ix), np.int32)
                rc = orc.svt_oracle_coeff_rate_batch(vp(q_h), vp(r), len(ix), vp(rtab_c), vp(rscan), vp(o))
                assert rc == 0
                bits[ix] = o
            t0 = time.perf_counter()
            list(ex.map(rate_chunk, range(len(chunks))))
            t["rate"] = time.perf_counter() - t0
            # deblocking: one picture per task (SB raster order inside a picture is serial in the reference's C path too)

            def lf_one(i):
                base = rec_h[i - 1]
                yd = B.YuvPlanes()
                yd.y, yd.u, yd.v = base.ctypes.data, base.ctypes.data + Hd * plane_w, base.ctypes.data + Hd * plane_w + Wd // 2
                yd.y_stride, yd.uv_stride, yd.width, yd.height = plane_w, plane_w, Wd, Hd
                lfm = np.ascontiguousarray(lfms[i - 1])
                rc = orc.svt_oracle_lf_frame(C.byref(yd), vp(lfm), lfm.shape[1], C.byref(thr), mi_rows, mi_cols, 0)
                assert rc == 0
            t0 = time.perf_counter()
            list(ex.map(lf_one, sample))
            t["lf"] = time.perf_counter() - t0
        return t

    t_all = run(ncpu)
    t_8 = run(8) if ncpu > 8 else t_all
    fps = lambda t: len(sample) / sum(t.values())
    return {"value": round(fps(t_all), 3), "unit": "frames/s", "cores": ncpu, "kind": "port", "cpu_model": cpu_model(),
            "value_8_cores": round(fps(t_8), 3),
            "stage_seconds_all_cores": {k: round(v, 3) for k, v in t_all.items()},
            "stage_seconds_8_cores": {k: round(v, 3) for k, v in t_8.items()},
            "sample": f"oracle (C restatement of the reference's C path, gcc -O3 -march=native on this host = auto-vectorised \"AVX2-class\" "
                      f"proxy) on {len(sample)} whole {Wd}x{Hd} pictures of the same mini-GOP (positions {sample}) through all six stages, "
                      f"threads over independent units (ME: SB ranges, transform / rate: block ranges, prediction / deblocking / analysis: pictures); "
                      f"wall-clock with {ncpu} threads (value) and with 8 (value_8_cores)"}


if __name__ == "__main__":
    main()
