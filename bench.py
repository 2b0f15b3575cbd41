#!/usr/bin/env python3
"""bench.py -- hot-path throughput of the MI355X implementation of SVT-VP9's block-level DSP path.

One "step" = one pass of the hot path over one mini-GOP (16 pictures of the 5-temporal-layer random-access
structure the reference uses at hierarchical_levels=4) of synthetic 3840x2160 8-bit 4:2:0 input, at the
enc-mode 8 / tune 1 (OQ) settings: every stage that is implemented runs for every picture
(motion estimation for the 16 inter pictures; transform/quant/recon and deblocking when built).
Inputs are resident in HBM before the timed region.  value = pictures / second (whole job, all ranks).

Contract: python bench.py --gpus N --steps K --warmup W ; for N>1 launched by torch.distributed.run, one
rank per GPU; GOP segments are independent (closed GOPs) so ranks share nothing and the
only collectives are the timing barrier / max-reduce.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "tests"))

W4K, H4K = 3840, 2160
MINIGOP = 16
# temporal layer of picture i (1..16) inside a 16-picture mini-GOP (5 layers, hierarchical_levels = 4)
LAYER = [4, 3, 4, 2, 4, 3, 4, 1, 4, 3, 4, 2, 4, 3, 4, 0]


def algorithmic_bytes_me(width, height, n_lists, l1_on):
    """src luma + src 1/16 (+1/4) + per list ref luma + ref 1/16 (+1/4) + results."""
    L = width * height
    nsb = ((width + 63) // 64) * ((height + 63) // 64)
    b = L * (1 + 1 / 16) + n_lists * L * (1 + 1 / 16) + 3400 * nsb
    if l1_on:
        b += (1 + n_lists) * L / 4
    return int(b)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--width", type=int, default=W4K)
    ap.add_argument("--height", type=int, default=H4K)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--stages", default="me,tq,lf", help="profiling aid: run only these stages (the contract run uses all three)")
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
    # one context per pipeline stage, each on its own torch stream so that torch events can bracket that stage's
    # launches inside the timed region
    # ME is the long pole of the step: its stream gets the higher priority so that freed CU slots go to it first
    prio = [int(x) for x in os.environ.get("SVT_BENCH_PRIO", "-1,0,0").split(",")]
    streams = [torch.cuda.Stream(device=local_rank, priority=prio[i]) for i in range(3)]
    # ME of different pictures is independent (it references source pictures, not reconstructions): the launch of the
    # deepest temporal layer and the launches of the other layers go to two ME streams so that the tail of one launch
    # is filled by the other
    n_me_streams = max(1, int(os.environ.get("SVT_BENCH_ME_STREAMS", "2")))
    for _ in range(n_me_streams - 1):
        streams.append(torch.cuda.Stream(device=local_rank, priority=prio[0]))
    # optional: disjoint CU sets for the stages (SVT_BENCH_CU_SPLIT = CUs out of every 4 that ME gets, e.g. "3"): the contexts
    # then own CU-masked streams and torch wraps them for events / cross-stream waits
    cu_env = os.environ.get("SVT_BENCH_CU_SPLIT", "0").split(",")
    cu_split = int(cu_env[0])
    rest_from = int(cu_env[1]) if len(cu_env) > 1 else cu_split   # "4,2": ME everywhere, TQ / LF on the upper half of each XCD
    ctxs = []
    if cu_split:
        n_cu = torch.cuda.get_device_properties(local_rank).multi_processor_count
        words = (n_cu + 31) // 32

        def mask(pred):
            m = (C.c_uint32 * words)()
            for i in range(n_cu):
                if pred((i // 8) % 4):   # a quarter granule of every XCD whichever way the driver enumerates the CUs
                    m[i // 32] |= 1 << (i % 32)
            return m

        m_me, m_rest = mask(lambda q: q < cu_split), mask(lambda q: q >= rest_from)
        for k_ in range(len(streams)):
            c_ = C.c_void_p()
            B.check(lib.svt_hip_ctx_create_cu_mask(C.byref(c_), local_rank, m_me if k_ in (0, 3) or k_ > 3 else m_rest, words))
            ctxs.append(c_)
            streams[k_] = torch.cuda.ExternalStream(lib.svt_hip_ctx_stream(c_), device=local_rank)
    else:
        for st_ in streams:
            c_ = C.c_void_p()
            B.check(lib.svt_hip_ctx_create_on_stream(C.byref(c_), local_rank, C.c_void_p(st_.cuda_stream)))
            ctxs.append(c_)
    ctx = ctxs[0]

    Wd, Hd = args.width, args.height
    nsb = T.n_sb(Wd, Hd)
    preset_name = "c3_2160p_m8" if Wd * Hd > 1920 * 1080 else ("c2_1080p_m8" if Wd * Hd > 720 * 576 else "c1_360p_m9")

    # ---- synthetic mini-GOP (+ the previous base-layer picture), resident in HBM ----
    frames = T.gen_clip(Wd, Hd, MINIGOP + 1, seed=GS.gop_seed(11, rank))  # rank r encodes its own GOP segment(s)
    dev = torch.device("cuda", local_rank)
    keep = []  # keeps device tensors alive

    def to_dev(a):
        t = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        keep.append(t)
        return t

    class DevPic:
        def __init__(self, luma):
            pa = T.PaPic(luma)
            self.t = [to_dev(a) for a, _ in pa.planes()]
            self.pads = [p for _, p in pa.planes()]

        def desc(self):
            d = B.PaPicture()
            for name, t, pad in zip(("full", "quarter", "sixteenth"), self.t, self.pads):
                pl = B.Plane()
                pl.buf = t.data_ptr()
                pl.stride = t.shape[1]
                pl.origin_x = pl.origin_y = pad
                pl.width, pl.height = t.shape[1] - 2 * pad, t.shape[0] - 2 * pad
                setattr(d, name, pl)
            return d

    pics = [DevPic(f) for f in frames]  # index 0 = previous base picture, 1..16 = the mini-GOP
    results = [to_dev(np.zeros((nsb, 85 * 10), np.int32)) for _ in range(MINIGOP + 1)]

    # references inside the mini-GOP (display order): picture i at layer l is predicted from the nearest
    # lower-layer pictures on both sides; the base-layer picture (16) from the previous base picture (0).
    def refs(i):
        if i == MINIGOP:
            return 0, 0
        l = LAYER[i - 1]
        span = MINIGOP >> l
        return i - span, i + span

    # ---- stage 1: motion estimation, one batched launch per temporal layer (parameters differ per layer) ----
    ctx_me, ctx_tq, ctx_lf = ctxs[:3]
    me_ctxs = [ctx_me] + ctxs[3:]
    me_launches = []
    for layer in range(5):
        idx = [i for i in range(1, MINIGOP + 1) if LAYER[i - 1] == layer]
        p = MC.preset(preset_name, 2, layer, 4)
        p.same_ref_poc = 1 if layer == 0 else 0
        chunk = max(1, int(os.environ.get("SVT_BENCH_ME_CHUNK", "8")))   # pictures per launch (a layer may be split)
        for j0 in range(0, len(idx), chunk):
            sub = idx[j0:j0 + chunk]
            n = len(sub)
            cur = (B.PaPicture * n)(*[pics[i].desc() for i in sub])
            r0 = (B.PaPicture * n)(*[pics[refs(i)[0]].desc() for i in sub])
            r1 = (B.PaPicture * n)(*[pics[refs(i)[1]].desc() for i in sub])
            res = (C.c_void_p * n)(*[results[i].data_ptr() for i in sub])
            me_launches.append((n, cur, r0, r1, p, res))
    # launches -> ME streams: largest first onto the least loaded stream
    me_slot, load = [0] * len(me_launches), [0] * len(me_ctxs)
    for li in sorted(range(len(me_launches)), key=lambda j: -me_launches[j][0]):
        k_ = load.index(min(load))
        me_slot[li] = k_
        load[k_] += me_launches[li][0]

    # ---- stage 2: transform / quantisation / reconstruction of the encode pass: every sample of the 4:2:0 picture
    # is covered by exactly one transform block (sizes 4x4..32x32 mixed per 32x32 area, DCT/ADST types mixed) ----
    rng = np.random.default_rng(5)
    plane_w = Wd  # Y rows followed by U rows and V rows (half width, stored at stride Wd) in one buffer
    yuv_rows = Hd + Hd // 2
    iscan, offs = T.iscan_array()
    qtabs = np.array([T.quant_table(a, b) for a, b in ((40, 48), (44, 52), (36, 44))], dtype=B.QUANT_DTYPE)
    areas = [(0, 0, Wd // 32, Hd // 32)] + [(Hd, 0, Wd // 64, Hd // 64), (Hd, Wd // 2, Wd // 64, Hd // 64)]
    blist = []
    for (row0, col0, aw, ah) in areas:
        ts_area = rng.integers(0, 4, (ah, aw))
        for ay in range(ah):
            for ax in range(aw):
                ts = int(ts_area[ay, ax])
                n = T.TX_N[ts]
                k = 32 // n
                ys = (row0 + ay * 32 + np.arange(k) * n)[:, None].repeat(k, 1).ravel()
                xs = (col0 + ax * 32 + np.arange(k) * n)[None, :].repeat(k, 0).ravel()
                blist.append((ts, ys, xs))
    tq_arr = []
    counts = np.zeros(4, np.int32)
    pos = 0
    for ts in range(4):
        ys = np.concatenate([b[1] for b in blist if b[0] == ts])
        xs = np.concatenate([b[2] for b in blist if b[0] == ts])
        n = T.TX_N[ts]
        a = np.zeros(len(ys), dtype=B.TQ_BLOCK_DTYPE)
        off = (ys * plane_w + xs).astype(np.uint32)
        a["src_off"] = a["pred_off"] = a["recon_off"] = off
        a["coeff_off"] = pos + np.arange(len(ys), dtype=np.uint32) * (n * n)
        tt = rng.integers(0, 4, len(ys)) if ts < 3 else np.zeros(len(ys), int)
        a["iscan_off"] = np.array([offs[(ts, int(t))] for t in range(4)], dtype=np.uint32)[tt]
        a["src_stride"] = a["pred_stride"] = a["recon_stride"] = plane_w
        a["tx_size"], a["tx_type"], a["qtab"], a["do_recon"] = ts, tt, rng.integers(0, 3, len(ys)), 1
        pos += len(ys) * n * n
        counts[ts] = len(ys)
        tq_arr.append(a)
    tq_blocks = np.concatenate(tq_arr)   # one picture's blocks, grouped by transform size
    n_coeff = pos
    # the whole mini-GOP is one batch (like ME's per-layer and LF's per-GOP launches): the 16 pictures live in one
    # buffer per plane set, picture k at byte offset k * pic_bytes, and the block list holds every picture's blocks,
    # grouped by transform size across pictures -> 4 launches per step, each big enough to fill the GPU
    pic_bytes = yuv_rows * plane_w
    all_arr = []
    for ts in range(4):
        for k in range(MINIGOP):
            a = tq_arr[ts].copy()
            for f in ("src_off", "pred_off", "recon_off"):
                a[f] += np.uint32(k * pic_bytes)
            a["coeff_off"] += np.uint32(k * n_coeff)
            all_arr.append(a)
    tq_blocks_all = np.concatenate(all_arr)
    d_blocks, d_qt, d_iscan = to_dev(tq_blocks_all.view(np.uint8)), to_dev(qtabs.view(np.uint8)), to_dev(iscan)
    d_q, d_dq = to_dev(np.zeros(MINIGOP * n_coeff, np.int16)), to_dev(np.zeros(MINIGOP * n_coeff, np.int16))
    d_eob = to_dev(np.zeros(len(tq_blocks_all), np.uint16))
    src_all = np.zeros((MINIGOP, yuv_rows, plane_w), np.uint8)
    pred_all = np.zeros_like(src_all)
    for i in range(1, MINIGOP + 1):
        y = frames[i]
        yuv = src_all[i - 1]
        yuv[:Hd] = y
        yuv[Hd:, :Wd // 2] = y[::2, ::2] // 2 + 32
        yuv[Hd:, Wd // 2:] = 128
        pred_all[i - 1] = np.clip(np.roll(yuv, (1, 2), (0, 1)).astype(np.int16) + rng.integers(-6, 7, yuv.shape, dtype=np.int16), 0, 255).astype(np.uint8)
    d_src, d_pred, d_rec = to_dev(src_all), to_dev(pred_all), to_dev(np.zeros_like(src_all))
    cnt_c = (C.c_int32 * 4)(*[int(v) * MINIGOP for v in counts.tolist()])

    # ---- stage 3: in-loop deblocking of the 16 reconstructed pictures, one batched launch ----
    mi_rows, mi_cols = Hd // 8, Wd // 8
    lfm = T.gen_lf_masks(np.random.default_rng(9), (mi_rows + 7) // 8, (mi_cols + 7) // 8)
    d_lfm = to_dev(np.ascontiguousarray(lfm).view(np.uint8))
    thr = B.LfThresh()
    lib.svt_hip_lf_thresh_init(C.byref(thr), 0)
    lf_desc = (B.YuvPlanes * MINIGOP)()
    for k in range(MINIGOP):
        base = d_rec.data_ptr() + k * pic_bytes
        d = lf_desc[k]
        d.y, d.u, d.v = base, base + Hd * plane_w, base + Hd * plane_w + Wd // 2
        d.y_stride, d.uv_stride, d.width, d.height = plane_w, plane_w, Wd, Hd
    lfm_ptrs = (C.c_void_p * MINIGOP)(*[d_lfm.data_ptr()] * MINIGOP)
    i32 = lambda v: (C.c_int32 * MINIGOP)(*[v] * MINIGOP)
    lfs, mrs, mcs = i32(lfm.shape[1]), i32(mi_rows), i32(mi_cols)

    def run_me():
        for st_ in streams[3:]:
            st_.wait_stream(streams[0])
        for li, (n, cur, r0, r1, p, res) in enumerate(me_launches):
            B.check(lib.svt_hip_me_batch_device(me_ctxs[me_slot[li]], n, cur, r0, r1, C.byref(p), res, None))
        for st_ in streams[3:]:
            streams[0].wait_stream(st_)

    def run_tq():
        B.check(lib.svt_hip_tq_batch_device(ctx_tq, C.c_void_p(d_src.data_ptr()), C.c_void_p(d_pred.data_ptr()), C.c_void_p(d_rec.data_ptr()),
                                            C.c_void_p(d_blocks.data_ptr()), cnt_c, C.c_void_p(d_qt.data_ptr()),
                                            C.c_void_p(d_iscan.data_ptr()), C.c_void_p(d_q.data_ptr()), C.c_void_p(d_dq.data_ptr()),
                                            C.c_void_p(d_eob.data_ptr())))

    def run_lf():
        B.check(lib.svt_hip_lf_batch_device(ctx_lf, MINIGOP, lf_desc, lfm_ptrs, lfs, C.byref(thr), mrs, mcs, 0))

    stages = set(args.stages.split(","))
    ev = []  # (stage, start event, stop event) of every stage of every timed step

    def staged(k, fn, record):
        if not record:
            return fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(streams[k])
        fn()
        e1.record(streams[k])
        ev.append((k, e0, e1))

    def step(record=False):
        # the three stages work on different pictures of the pipeline (as the reference's ME / EncDec threads do), so
        # they are issued on three streams; deblocking of a step's pictures is ordered after their reconstruction
        if "me" in stages:
            staged(0, run_me, record)
        if "tq" in stages:
            staged(1, run_tq, record)
        streams[2].wait_stream(streams[1])
        if "lf" in stages:
            staged(2, run_lf, record)

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
    for _ in range(args.steps):
        step(record=True)
    sync()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    dt = GS.reduce_elapsed(dt, dist if world > 1 else None, dev)

    # ---- per-kernel time: HIP events on each stage's own stream, bracketing that stage's launches of every timed
    # step (so it includes whatever slowdown the overlap with the other stages causes, like a rocprofv3 trace) ----
    stage_ms = [0.0, 0.0, 0.0]
    for k, e0, e1 in ev:
        stage_ms[k] += e0.elapsed_time(e1)
    me_ms, tq_ms, lf_ms = [v / args.steps for v in stage_ms]
    l1_on = bool(me_launches[0][4].enable_hme_level_1_flag)
    me_bytes = MINIGOP * algorithmic_bytes_me(Wd, Hd, 2, l1_on)
    L = Wd * Hd
    tq_bytes = MINIGOP * int(7.5 * L)                      # src 1.5L + pred 1.5L + qcoeff 3L + recon 1.5L (the kernel also writes dqcoeff, +3L)
    lf_bytes = MINIGOP * (3 * L + 160 * nsb)               # recon read + write (3L) + masks
    me_ms, tq_ms, lf_ms = max(me_ms, 1e-9), max(tq_ms, 1e-9), max(lf_ms, 1e-9)
    achieved = me_bytes / (me_ms * 1e-3) / 1e9  # GB/s

    # ---- picture-analysis pre-ME stage (row f-1, not part of `value`): rebuild the three padded planes of the 16
    # pictures from their luma in one batched launch, timed on its own with HIP events ----
    d_lumas = [to_dev(frames[i]) for i in range(1, MINIGOP + 1)]
    pa_out = (B.PaPicture * MINIGOP)(*[pics[i].desc() for i in range(1, MINIGOP + 1)])
    pa_ptrs = (C.c_void_p * MINIGOP)(*[t.data_ptr() for t in d_lumas])
    pa_strides = (C.c_int32 * MINIGOP)(*[Wd] * MINIGOP)
    pa_ms = 0.0
    for rep in range(4):
        B.check(lib.svt_hip_pa_prepare_batch_device(ctx_me, MINIGOP, pa_ptrs, pa_strides, pa_out, 1 if l1_on else 0))
        B.check(lib.svt_hip_ctx_synchronize(ctx_me))
        if rep:
            pa_ms += lib.svt_hip_last_kernel_ms(ctx_me) / 3
    pa_bytes = MINIGOP * int(Wd * Hd + (Wd + 136) * (Hd + 136) + (Wd // 4 + 32) * (Hd // 4 + 32) + (((Wd // 2 + 64) * (Hd // 2 + 64)) if l1_on else 0))

    # ---- inter prediction (row f-2, not part of `value`): 8-tap motion compensation of the 16 pictures from a mode-info
    # grid (64..8 partitions, ~40 % compound, sub-sample MVs), two padded reference pictures, one batched launch ----
    mcase = T.make_mc_case(3, width=Wd, height=Hd, mv_range=24, intra_share=0.1)
    mc_keep = [to_dev(np.ascontiguousarray(mcase["mi"]).view(np.uint8))]
    mc_refs = []
    for (y, u, v) in mcase["refs"]:
        ty, tu, tv = to_dev(y), to_dev(u), to_dev(v)
        mc_keep += [ty, tu, tv]
        pad = mcase["pad"]
        mc_refs.append((ty.data_ptr() + pad * y.shape[1] + pad, tu.data_ptr() + (pad // 2) * u.shape[1] + pad // 2,
                        tv.data_ptr() + (pad // 2) * v.shape[1] + pad // 2, y.shape[1], u.shape[1]))
    d_mcpred = to_dev(np.zeros((MINIGOP, Hd * 3 // 2, Wd), np.uint8))
    mc_pics = (B.McPicture * MINIGOP)()
    for k in range(MINIGOP):
        mp = mc_pics[k]
        mp.d_mi, mp.mi_stride, mp.mi_rows, mp.mi_cols, mp.use_subpel = mc_keep[0].data_ptr(), mcase["mi_cols"], mcase["mi_rows"], mcase["mi_cols"], 1
        for l in range(2):
            r = mp.ref[l]
            r.y, r.u, r.v, r.y_stride, r.uv_stride = mc_refs[l]
            r.width, r.height = Wd, Hd
        base = d_mcpred.data_ptr() + k * (Hd * 3 // 2) * Wd
        mp.pred.y, mp.pred.u, mp.pred.v = base, base + Hd * Wd, base + Hd * Wd + (Hd // 2) * (Wd // 2)
        mp.pred.y_stride, mp.pred.uv_stride, mp.pred.width, mp.pred.height = Wd, Wd // 2, Wd, Hd
    mc_ms = 0.0
    for rep in range(4):
        B.check(lib.svt_hip_inter_pred_batch_device(ctx_me, MINIGOP, mc_pics))
        B.check(lib.svt_hip_ctx_synchronize(ctx_me))
        if rep:
            mc_ms += lib.svt_hip_last_kernel_ms(ctx_me) / 3
    inter = mcase["mi"]["ref_list"][:, :, 0] >= 0
    comp = mcase["mi"]["ref_list"][:, :, 1] >= 0
    # per 8x8 unit: 96 bytes (64 luma + 2 x 16 chroma) read per reference and written once, + the 12-byte mode-info record
    mc_bytes = MINIGOP * int(96 * (inter.sum() + (inter & comp).sum()) + 96 * inter.sum() + 12 * inter.size)

    # ---- coefficient rate estimation (row f-4, not part of `value`): bits of every transform block of the mini-GOP from
    # the quantised coefficients the TQ stage just wrote (d_q), one launch ----
    rtab, rscan = T.rate_tables()
    roffs, _ = T.rate_scan_offsets()
    rb = np.zeros(len(tq_blocks_all), dtype=B.RATE_BLOCK_DTYPE)
    rb["coeff_off"] = tq_blocks_all["coeff_off"]
    rb["tx_size"] = tq_blocks_all["tx_size"]
    rb["scan_off"] = np.array([roffs[(int(a), int(b) if a < 3 else 0)] for a, b in zip(tq_blocks_all["tx_size"], tq_blocks_all["tx_type"])], np.uint32)
    rb["plane_type"] = (tq_blocks_all["src_off"] % pic_bytes >= Hd * plane_w).astype(np.uint8)
    rb["is_inter"] = 0     # the bench's TQ blocks carry intra tx types (ADST mixes), so they are costed as intra blocks
    rb["ctx"] = np.random.default_rng(8).integers(0, 3, len(rb))
    rb["eob"] = d_eob.cpu().numpy().view(np.uint16)[:len(rb)]
    d_rb, d_rt, d_rs = to_dev(rb.view(np.uint8)), to_dev(np.ascontiguousarray(rtab).reshape(1).view(np.uint8)), to_dev(rscan)
    d_bits = to_dev(np.zeros(len(rb), np.int32))
    rate_ms = 0.0
    for rep in range(4):
        B.check(lib.svt_hip_coeff_rate_batch_device(ctx_me, C.c_void_p(d_q.data_ptr()), C.c_void_p(d_rb.data_ptr()), len(rb), C.c_void_p(d_rt.data_ptr()),
                                                    C.c_void_p(d_rs.data_ptr()), C.c_void_p(d_bits.data_ptr())))
        B.check(lib.svt_hip_ctx_synchronize(ctx_me))
        if rep:
            rate_ms += lib.svt_hip_last_kernel_ms(ctx_me) / 3
    # per block: the n*n coefficients up to eob are read (2 bytes each, whole 64-byte lines), 16-byte descriptor, 4-byte result
    rate_bytes = int(np.sum(np.minimum((16 << (2 * rb["tx_size"].astype(np.int64))), ((rb["eob"].astype(np.int64) * 2 + 63) // 64 + 1) * 32)) * 2 + 20 * len(rb))

    if rank != 0:
        return
    # HBM traffic of the dominant kernel per step, from the committed rocprofv3 PMC passes (profiles/r01_pmc_traffic.md)
    traffic = None
    tj = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tj) and (Wd, Hd) == (W4K, H4K):
        traffic = json.load(open(tj)).get("svt_me_sb_kernel", {}).get("bytes_per_step")
    valu = None
    if os.path.exists(tj) and (Wd, Hd) == (W4K, H4K):
        vi = json.load(open(tj)).get("svt_me_sb_kernel", {}).get("valu_wave_insts_per_step")
        if vi:
            # the kernel's real roof: 64-lane VALU instructions issued (rocprofv3 SQ_INSTS_VALU, profiles/) per second against
            # 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz (MI355X_MICROARCH.md)
            ach = vi * 64 / (me_ms * 1e-3) / 1e12
            valu = {"achieved": round(ach, 2), "peak": 39.3, "unit": "T lane-ops/s", "frac": round(ach / 39.3, 4), "wave_insts_per_step": vi}
    fps = GS.aggregate_rate(MINIGOP, args.steps, world, dt)
    out = {
        "metric": "encoded frames/sec (block-level DSP hot path: ME + DCT/quant/recon + deblock), 4Kp60 yuv420p enc-mode 8",
        "value": round(fps, 2),
        "unit": "frames/s",
        "mpixels_per_s": round(fps * Wd * Hd / 1e6, 1),
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u8",
        "data": "synthetic",
        "config": {"workload": f"{Wd}x{Hd} 8-bit yuv420p, -enc-mode 8 -tune 1, 1xMI355X per rank; step = one 16-picture "
                               "mini-GOP (5 temporal layers, B pictures, 2 reference lists) through the hot path",
                   "stages": ["motion_estimation", "transform_quant_recon", "deblocking"], "pictures_per_step": MINIGOP,
                   "parallelism": f"gop-shard x{world}"},
        "roofline": {"bound": "hbm", "kernel": "svt_me_sb_kernel", "achieved": round(achieved, 2), "peak": 8000.0,
                     "unit": "GB/s", "frac": round(achieved / 8000.0, 5), "traffic": traffic,
                     "launches_per_step": len(me_launches),
                     "kernel_ms_per_step": round(me_ms, 3), "algorithmic_bytes_per_step": me_bytes, "valu": valu},
        "kernels": {k: {"ms_per_step": round(ms, 3), "algorithmic_bytes_per_step": b, "GB_per_s": round(b / (ms * 1e-3) / 1e9, 2),
                        "frac_of_8TBps": round(b / (ms * 1e-3) / 8e12, 5)}
                    for k, ms, b in (("svt_me_sb_kernel", me_ms, me_bytes), ("svt_tq_kernel<4|8|16|32>", tq_ms, tq_bytes),
                                     ("svt_lf_kernel", lf_ms, lf_bytes), ("svt_pa_plane_kernel (pre-ME stage, outside value)", pa_ms, pa_bytes),
                                     ("svt_mc_kernel (inter prediction in front of TQ, outside value)", max(mc_ms, 1e-9), mc_bytes),
                                     ("svt_rate_kernel (coefficient rate after TQ, outside value)", max(rate_ms, 1e-9), rate_bytes))},
    }
    if not args.no_cpu_baseline:
        # oracle (scalar C restatement of the reference C path), single thread, on a bounded sample of the same
        # workload: whole pictures of the mini-GOP through each of the three stages (about 10-20 s of CPU time)
        orc = T.oracle()
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        n_me, t_me, used = 0, 0.0, []
        for i in (8, 4, 2, 1, 12, 6):
            l = LAYER[i - 1]
            a, b = refs(i)
            cur, r0, r1 = T.PaPic(frames[i]), T.PaPic(frames[a]), T.PaPic(frames[b])
            p = MC.preset(preset_name, 2, l, 4)
            t1 = time.perf_counter()
            T.oracle_me_picture(cur, r0, r1, p)
            t_me += time.perf_counter() - t1
            n_me += 1
            used.append(i)
            if t_me >= 6.0:
                break
        src_h, pred_h = src_all[0], pred_all[0]
        rec_h = np.zeros_like(src_h)
        q_h, dq_h, eob_h = np.zeros(n_coeff, np.int16), np.zeros(n_coeff, np.int16), np.zeros(len(tq_blocks), np.uint16)
        t1 = time.perf_counter()
        rc = orc.svt_oracle_tq_batch(vp(src_h), vp(pred_h), vp(rec_h), vp(tq_blocks), len(tq_blocks), vp(qtabs), vp(iscan), vp(q_h),
                                     vp(dq_h), vp(eob_h))
        t_tq = time.perf_counter() - t1
        assert rc == 0
        yd = B.YuvPlanes()
        yd.y, yd.u, yd.v = rec_h.ctypes.data, rec_h.ctypes.data + Hd * plane_w, rec_h.ctypes.data + Hd * plane_w + Wd // 2
        yd.y_stride, yd.uv_stride, yd.width, yd.height = plane_w, plane_w, Wd, Hd
        lfm_h = np.ascontiguousarray(lfm)
        t1 = time.perf_counter()
        rc = orc.svt_oracle_lf_frame(C.byref(yd), vp(lfm_h), lfm_h.shape[1], C.byref(thr), mi_rows, mi_cols, 0)
        t_lf = time.perf_counter() - t1
        assert rc == 0
        per_pic = t_me / n_me + t_tq + t_lf
        out["cpu_baseline"] = {"value": round(1.0 / per_pic, 4), "unit": "frames/s", "cores": 1, "kind": "port",
                               "sample": f"oracle (scalar C restatement, gcc -O2, one thread) on whole {Wd}x{Hd} pictures: ME of "
                                         f"{n_me} B pictures (mini-GOP positions {used}) {t_me / n_me:.2f} s/picture, transform/quant/"
                                         f"recon of 1 picture {t_tq:.2f} s, deblocking of 1 picture {t_lf:.2f} s"}
    print(json.dumps(out))
    for c_ in ctxs:
        lib.svt_hip_ctx_destroy(c_)


if __name__ == "__main__":
    main()
