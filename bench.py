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
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--width", type=int, default=W4K)
    ap.add_argument("--height", type=int, default=H4K)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl")

    import me_configs as MC
    import svt_testlib as T
    B = T.B
    lib = B.load()
    ctx = C.c_void_p()
    B.check(lib.svt_hip_ctx_create(C.byref(ctx), local_rank))

    Wd, Hd = args.width, args.height
    nsb = T.n_sb(Wd, Hd)
    preset_name = "c3_2160p_m8" if Wd * Hd > 1920 * 1080 else ("c2_1080p_m8" if Wd * Hd > 720 * 576 else "c1_360p_m9")

    # ---- synthetic mini-GOP (+ the previous base-layer picture), resident in HBM ----
    frames = T.gen_clip(Wd, Hd, MINIGOP + 1, seed=11 + rank)
    dev = torch.device("cuda", local_rank)

    class DevPic:
        def __init__(self, luma):
            pa = T.PaPic(luma)
            self.t = [torch.from_numpy(a).to(dev) for a, _ in pa.planes()]
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
    results = [torch.zeros((nsb, 85 * 10), dtype=torch.int32, device=dev) for _ in range(MINIGOP + 1)]

    # references inside the mini-GOP (display order): picture i at layer l is predicted from the nearest
    # lower-layer pictures on both sides; the base-layer picture (16) from the previous base picture (0).
    def refs(i):
        if i == MINIGOP:
            return 0, 0
        l = LAYER[i - 1]
        span = MINIGOP >> l
        return i - span, i + span

    # group pictures by temporal layer: one batched launch per layer (parameters differ per layer)
    launches = []
    for layer in range(5):
        idx = [i for i in range(1, MINIGOP + 1) if LAYER[i - 1] == layer]
        nl = 2
        p = MC.preset(preset_name, nl, layer, 4)
        p.same_ref_poc = 1 if layer == 0 else 0
        n = len(idx)
        cur = (B.PaPicture * n)(*[pics[i].desc() for i in idx])
        r0 = (B.PaPicture * n)(*[pics[refs(i)[0]].desc() for i in idx])
        r1 = (B.PaPicture * n)(*[pics[refs(i)[1]].desc() for i in idx])
        res = (C.c_void_p * n)(*[results[i].data_ptr() for i in idx])
        launches.append((n, cur, r0, r1, p, res))

    def step():
        for n, cur, r0, r1, p, res in launches:
            B.check(lib.svt_hip_me_batch_device(ctx, n, cur, r0, r1, C.byref(p), res, None))

    def sync():
        B.check(lib.svt_hip_ctx_synchronize(ctx))
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # ---- per-kernel timing (HIP events on the context's stream), outside the timed region ----
    kern_ms, kern_bytes = 0.0, 0
    l1_on = bool(launches[0][4].enable_hme_level_1_flag)
    reps = 5
    for n, cur, r0, r1, p, res in launches:
        for _ in range(reps):
            B.check(lib.svt_hip_me_batch_device(ctx, n, cur, r0, r1, C.byref(p), res, None))
            B.check(lib.svt_hip_ctx_synchronize(ctx))
            kern_ms += lib.svt_hip_last_kernel_ms(ctx)
        kern_bytes += n * algorithmic_bytes_me(Wd, Hd, 2, l1_on)
    kern_ms /= reps
    achieved = kern_bytes / (kern_ms * 1e-3) / 1e9  # GB/s

    if rank != 0:
        return
    fps = MINIGOP * args.steps * world / dt
    out = {
        "metric": "encoded frames/sec (block-level DSP hot path: motion estimation), 4Kp60 yuv420p enc-mode 8",
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
                   "stages": ["motion_estimation"], "pictures_per_step": MINIGOP, "parallelism": f"gop-shard x{world}"},
        "roofline": {"bound": "hbm", "kernel": "svt_me_sb_kernel", "achieved": round(achieved, 2), "peak": 8000.0,
                     "unit": "GB/s", "frac": round(achieved / 8000.0, 5), "traffic": None,
                     "kernel_ms_per_step": round(kern_ms, 3), "algorithmic_bytes_per_step": kern_bytes},
    }
    if not args.no_cpu_baseline:
        # oracle (scalar C restatement of the reference C path), single thread, on a bounded sample of the same
        # workload: whole B pictures of the mini-GOP until >= 10 s of CPU time has been spent
        n_done, cdt, used = 0, 0.0, []
        for i in (8, 4, 2, 1, 12, 6, 3, 5):
            l = LAYER[i - 1]
            a, b = refs(i)
            cur, r0, r1 = T.PaPic(frames[i]), T.PaPic(frames[a]), T.PaPic(frames[b])
            p = MC.preset(preset_name, 2, l, 4)
            t1 = time.perf_counter()
            T.oracle_me_picture(cur, r0, r1, p)
            cdt += time.perf_counter() - t1
            n_done += 1
            used.append(i)
            if cdt >= 10.0:
                break
        out["cpu_baseline"] = {"value": round(n_done / cdt, 4), "unit": "frames/s", "cores": 1, "kind": "port",
                               "sample": f"oracle ME (scalar C restatement, gcc -O2) on {n_done} whole {Wd}x{Hd} B pictures "
                                         f"(mini-GOP positions {used}) in {cdt:.1f} s, one thread"}
    print(json.dumps(out))
    lib.svt_hip_ctx_destroy(ctx)


if __name__ == "__main__":
    main()
