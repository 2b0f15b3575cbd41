#!/usr/bin/env python3
"""bench.py -- hot-path throughput of the MI355X implementation of SVT-VP9's block-level DSP path, dependency-true.

One "step" = one mini-GOP (16 pictures, 5 temporal layers with tune 1 / 8 pictures, 4 layers with tune 0; B pictures with 2
reference lists) of EACH of G concurrent closed GOPs (closed GOPs are independent streams of work) of synthetic
8-bit 4:2:0 input through the whole data path in the order the data dependencies of the encoder impose:

    ME side (source pictures only, one mini-GOP ahead, own streams):
        picture analysis (padded + decimated planes from the source luma)  ->  motion estimation (one launch per temporal layer)
    EncDec side, per batch of mutually independent pictures, ONE call of the picture-level driver (svt_hip_encdec_batch_device):
        inter prediction FROM THE DEBLOCKED, PADDED RECONSTRUCTION of the reference pictures  ->  transform-block lists built on the
        device from the mode-info grid  ->  residual / transform / quantisation / reconstruction into the picture's reference buffer
        ->  eob map + skip flags  ->  loop-filter masks (device)  ->  in-loop deblocking in place  ->  border padding in place:
        the picture is now a reference picture

Nothing between mode decision's output (the mode-info grid) and the padded reference picture is prepared outside the timed region:
block lists, skip flags and filter masks are derived on the device inside it.  The mode-info grids themselves are mode decision's
output -- host control logic outside this path -- and are synthesised ONCE before the timed loop from a first ME pass.
Which pictures form a batch is the schedule: "diagonal" (per step temporal layer l of the mini-GOP l steps back: every picture is
predicted from reconstructions finished in earlier steps) or "waves" (the layers of the newest mini-GOP one after the other).
`value`: every picture reconstructed and deblocked (the reference's behaviour with reconstructed output enabled);
`value_reference_flags`: the per-picture stage flags the reference derives WITHOUT reconstructed output at this preset
(svt_hip_encdec_flags_derive: at enc-mode 8 deblocking on base-layer pictures only, the deepest layer not reconstructed).
Inputs are resident in HBM before the timed region.  value = pictures / second (whole job, all ranks).

Contract: python bench.py --gpus N --steps K --warmup W ; for N>1 launched by torch.distributed.run, one rank per GPU;
GOP segments are independent (closed GOPs) so ranks share nothing and the only collectives are the
timing barrier / max-reduce.  --preset c1 | c2 | c3 | c5 selects the BASELINE.json configuration (default c3, the one `metric` is quoted on).
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
PRESETS = {   # BASELINE.json configs: (width, height, enc_mode, tune)
    "c1": (640, 360, 9, 1),    # the reference's own CPU-runnable case (its parameters; there is no CPU path here)
    "c2": (1920, 1080, 8, 1),
    "c3": (W4K, H4K, 8, 1),
    "c5": (W4K, H4K, 3, 0),
}
Q_INDEX = 160   # -q 40: quantizer_to_qindex[40]
PAD = 80        # border of a reference picture: 64 + 16 (Codec/EbEncHandle.c:968-971); chroma half of it
MINIGOP = 16
LEVELS = 4
# temporal layer of picture i (1..MINIGOP) inside a mini-GOP
LAYER = [4, 3, 4, 2, 4, 3, 4, 1, 4, 3, 4, 2, 4, 3, 4, 0]
STAGES = ("pa", "me", "mc", "lists", "tq", "skip", "lf", "pad")
ED_STAGE_NAMES = ("mc", "lists", "tq", "skip", "lf", "pad")   # SVT_ENCDEC_STAGE_* of svtvp9_hip.h
ED_BATCH = 32   # pictures per driver call at most (svt_hip_encdec_batch_device)


def set_structure(levels):
    """hierarchical_levels of the prediction structure: 4 (tune != 0 under CQP) or 3 (Codec/EbEncHandle.c:2167-2175)"""
    global MINIGOP, LEVELS, LAYER
    LEVELS, MINIGOP = levels, 1 << levels
    LAYER = [0 if i == MINIGOP else levels - ((i & -i).bit_length() - 1) for i in range(1, MINIGOP + 1)]


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
    pictures on both sides; the base-layer picture (MINIGOP) from the previous base picture (0)"""
    if i == MINIGOP:
        return 0, 0
    span = MINIGOP >> LAYER[i - 1]
    return i - span, i + span


def pics_of_layer(layer):
    return [i for i in range(1, MINIGOP + 1) if LAYER[i - 1] == layer]


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


def build_lf_mode_info(B, k_cell, mi_rows, mi_cols, level):
    """the other half of the same decision (svt_lf_mode_info): block size, transform size, filter level; `skip` is the transform
    stage's output and is written by the driver"""
    lmi = np.zeros((mi_rows, mi_cols), dtype=B.LF_MODE_INFO_DTYPE)
    lmi["sb_type"] = np.where(k_cell == 3, 9, np.where(k_cell == 2, 6, 3))
    lmi["tx_size"], lmi["is_inter"], lmi["filter_level"] = k_cell, 1, level
    return lmi


def partition_kinds(rng, Wd, Hd):
    area_rows, area_cols = (Hd + 31) // 32, (Wd + 31) // 32
    kinds = rng.integers(0, 4, (area_rows, area_cols))
    if Hd % 32:   # blocks must not reach below the picture
        kinds[-1] = np.minimum(kinds[-1], 2 if Hd % 32 == 16 else 1)
    if Wd % 32:
        kinds[:, -1] = np.minimum(kinds[:, -1], 2 if Wd % 32 == 16 else 1)
    return kinds


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def effective_cpus():
    """host cores this process may actually use: the hardware thread count, capped by the container's CPU-time quota (cgroup v2
    cpu.max / v1 cfs quota) -- on the GPU boxes of this pool 256 hardware threads are visible but the quota is 16 CPUs' worth of time,
    so more than 16 busy processes only take turns"""
    n = os.cpu_count() or 1
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = int(q) / int(per)
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    return max(1, min(n, int(quota))) if quota else n, n, quota


def build_native_oracle():
    """oracle/*.c compiled for this host (auto-vectorised SAD / transform loops: the "AVX2-class" CPU proxy of SURVEY 8(d))"""
    out = os.path.join(tempfile.gettempdir(), f"liboracle_native_{os.getuid()}.so")
    src = [os.path.join(ROOT, "oracle", f) for f in ("oracle_me.c", "oracle_tq.c", "oracle_lf.c", "oracle_pa.c", "oracle_mc.c", "oracle_rate.c")]
    subprocess.check_call(["gcc", "-std=gnu11", "-O3", "-march=native", "-fPIC", "-shared", "-Wno-unused-function", "-o", out] + src + ["-lpthread"])
    return out


def intra_pass_times():
    """The encode pass of a key frame (svt_hip_encdec_intra_device: wavefront prediction + transform, then deblocking and border) is not
    part of the step -- one picture per GOP, latency-bound, beside the inter batches.  tools/intra_time.py times it alone on one 2160p
    picture (inputs resident, wall clock around call + synchronisation) for uniform block sizes and a random partition."""
    exe = os.path.join(ROOT, "tools", "intra_time.py")
    if not os.path.exists(exe):
        return None
    r = subprocess.run([sys.executable, exe], capture_output=True, text=True)
    if r.returncode != 0:
        return {"error": (r.stdout + r.stderr).strip()[-200:]}
    ms = {}
    for line in r.stdout.splitlines():
        if line.rstrip().endswith(" ms"):
            name, val = line.rsplit(None, 2)[0].strip(), float(line.rsplit(None, 2)[1])
            ms[name] = val
    return {"ms_per_2160p_picture": ms, "what": "svt_hip_encdec_intra_device on one 3840x2160 picture, q index 140, deblocking + border included; persistent 64-lane "
                                              "workgroups (two per CU) draw (16x16 luma cell, plane) tickets in anti-diagonal order: 374 dependent cell steps per picture"}


def api_path_rate(frames, Wd, Hd, enc_mode, tune, n_send=130):
    """SURVEY 8(d)'s metric through the PUBLIC API (libSvtVp9Enc.so, the reference's eb_vp9_svt_* entry points): wall-clock from the
    first send_picture to the EOS packet, host buffers handed over (PCIe inside the clock: all three planes of every picture), measured
    by the plain-C caller app/svt_enc_api_bench.c in its own process.  Behind the API the library runs ALL the stages of this path:
    picture analysis -> motion estimation (one batched launch per mini-GOP) + per-SB statistics -> mode decision stand-in -> inter
    prediction from the reconstructed references -> transform / quantisation / reconstruction -> deblocking -> reference padding,
    one batch per temporal layer, with the per-picture stage flags the reference derives.  Two runs: recon_file = 0 (the reference's
    default: at enc-mode 8 only base-layer pictures are deblocked and the deepest layer is not reconstructed) and recon_file = 1 with
    every reconstructed picture fetched through eb_vp9_svt_get_recon (every picture reconstructed and deblocked, 12.4 MB per 4K
    picture back over PCIe).  n_send = 130: two closed GOPs of 65 pictures at 60 frames/s."""
    exe = os.path.join(ROOT, "app", "svt_enc_api_bench")
    if not os.path.exists(exe):
        return None
    out = {}
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "clip.yuv")
        with open(path, "wb") as f:
            for y in frames:
                y = np.ascontiguousarray(y)
                f.write(y.tobytes())
                f.write((y[::2, ::2] // 2 + 32).astype(np.uint8).tobytes())     # SURVEY 8(d)'s clip: U = Y / 2 + 32, V = 128
                f.write(np.full((Hd // 2, Wd // 2), 128, np.uint8).tobytes())
        def run_best(args, reps):
            """best of `reps` runs (a 130-picture run lasts 40 ms: one preempted copy thread shows); None + the error text when a run fails"""
            best, err = None, None
            for _ in range(reps):
                r = subprocess.run([exe, path, str(Wd), str(Hd), str(len(frames))] + args, capture_output=True, text=True)
                if r.returncode != 0:
                    err = f"svt_enc_api_bench rc={r.returncode}: {(r.stdout + r.stderr).strip()[-200:]}"
                    break
                d_ = json.loads(r.stdout.strip().splitlines()[-1])
                if best is None or d_["frames_per_s"] > best["frames_per_s"]:
                    best = d_
            return best, err
        for recon in (0, 1):
            d, err = run_best([str(n_send), str(enc_mode), str(tune), str(recon)], 3)
            if d is None:
                return {"error": err}
            out["with_recon_output" if recon else "default"] = {"value": d["frames_per_s"], "frames": d["frames"], "seconds": d["seconds"], "me_launches": d["me_launches"],
                                                                "mpixels_per_s": round(d["frames_per_s"] * Wd * Hd / 1e6, 1), "recon_pictures": d["recon_pictures"]}
        # the N-device host shape on ONE GPU: two contexts (two picture rings, two sets of streams, two feeder threads) take the closed GOPs
        # by turns -- exercises what `SVT_HIP_DEVICES=a,b` runs on a multi-GPU node; no scaling claim (one GPU's worth of compute and link)
        two, _ = run_best([str(n_send), str(enc_mode), str(tune), "0", "0,0"], 3)
        # the same path over 600 pictures (10 s of video: fill and drain, ~12 ms, no longer weigh) -- with and without the reconstructions fetched
        long_run = {}
        for recon in (0, 1):
            d, _ = run_best(["600", str(enc_mode), str(tune), str(recon)], 2)
            if d is not None:
                long_run["with_recon_output" if recon else "default"] = {"value": d["frames_per_s"], "frames": d["frames"], "seconds": d["seconds"]}
    d0 = out["default"]
    return {"value": d0["value"], "unit": "frames/s", "frames": d0["frames"], "seconds": d0["seconds"], "me_launches": d0["me_launches"], "mpixels_per_s": d0["mpixels_per_s"],
            "with_recon_output": out["with_recon_output"],
            "value_600_pictures": long_run.get("default", {}).get("value"), "long_run": long_run or None,
            "two_contexts_one_gpu": ({"value": two["frames_per_s"], "frames": two["frames"], "devices": "0,0",
                                      "note": "GOPs dealt to two contexts of the same GPU (the multi-device host shape: per-device ring, streams, feeder thread); not a scaling figure"}
                                     if two else None),
            "stages": ["pa", "me", "me_stats", "md_stand_in", "mc", "lists", "tq", "skip", "masks+lf", "pad"],
            "what": f"{Wd}x{Hd} -enc-mode {enc_mode} -tune {tune} -q 40, {d0['frames']} pictures through eb_vp9_svt_enc_send_picture / eb_vp9_svt_get_packet "
                    "(app/svt_enc_api_bench.c; best of three runs, two for the 600-picture ones): first send_picture -> EOS packet, host buffers in (Y, Cb, Cr), PCIe included; every stage of the path behind the "
                    "API, one mini-GOP at a time (uploads + picture analysis on an input stream, reconstruction copies on an output stream, beside the main stream), the stage flags per picture as the reference derives them (recon_file = 0: deblocking on base-layer pictures "
                    "only, no reconstruction of the deepest layer; with_recon_output: recon_file = 1, all pictures reconstructed + deblocked and fetched with "
                    "eb_vp9_svt_get_recon); zero-byte packets (no entropy coding); picture 0 is a key frame coded by the intra encode pass (svt_hip_encdec_intra_device, stand-in decision: 16x16 DC), ~4.3 ms of the run; value_600_pictures / long_run: the same over 600 pictures -- the host -> device link bounds it (DESIGN.md section 7)"}


def reference_me_rate(T, B, orc, frames, Wd, Hd, enc_mode, tune, l1_on, ncpu):
    """The REFERENCE's own motion_estimate_sb (oracle/_ref/ref_me_sb = Codec/EbMotionEstimation.c compiled from /root/reference in
    the build container, C path, gcc -O2; it travels to the GPU box as a prebuilt file) timed beside the port on the same
    picture: one B picture of temporal layer 2 of the mini-GOP, SB ranges over up to 64 processes / threads; the seconds
    are the SB loops' own (clock_gettime inside the harness: no request I/O).  Returns None when the prebuilt reference is
    absent or cannot run this preset (the SSD fractional search of enc-mode <= 4 calls a yasm-only symbol)."""
    from concurrent.futures import ThreadPoolExecutor
    import struct
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_me_sb")
    if not os.path.exists(exe) or enc_mode <= 4:
        return None
    i = MINIGOP // 4
    a, b = refs_of(i)
    pics = [T.PaPic(frames[j]) for j in (i, a, b)]
    p = B.me_params_preset(Wd, Hd, enc_mode, tune, 2, LAYER[i - 1], LEVELS)
    nsb = T.n_sb(Wd, Hd)
    workers = max(1, min(ncpu, 64))
    per = max(1, min(32, nsb // workers))       # bounded sample: at most 32 SBs per worker
    ranges = [(k * per, min(nsb, (k + 1) * per)) for k in range(workers) if k * per < nsb]
    with tempfile.TemporaryDirectory() as td:
        reqs = []
        for k, (s0, s1) in enumerate(ranges):
            rq = os.path.join(td, f"rq{k}")
            with open(rq, "wb") as f:
                f.write(struct.pack("<i", 0x454D5653))
                f.write(bytes(p))
                f.write(struct.pack("<ii", s0, s1))
                for pic in pics:
                    for arr, pd in pic.planes():
                        hh, ww = arr.shape
                        f.write(struct.pack("<6i", ww, pd, pd, ww - 2 * pd, hh - 2 * pd, arr.size))
                        f.write(arr.tobytes())
            reqs.append((rq, os.path.join(td, f"rs{k}")))

        def run_ref(k):
            subprocess.check_call([exe, reqs[k][0], reqs[k][1]])
            raw = open(reqs[k][1], "rb").read()
            return struct.unpack_from("<d", raw, len(raw) - 8)[0], np.frombuffer(raw, dtype=B.ME_RESULT_DTYPE, count=nsb * 85, offset=4).reshape(nsb, 85)
        with ThreadPoolExecutor(len(ranges)) as ex:
            ref_out = list(ex.map(run_ref, range(len(ranges))))
        one = run_ref(0)    # alone on the machine: the single-thread figure
    res = np.zeros((nsb, 85), dtype=B.ME_RESULT_DTYPE)
    dc, d0, d1 = (pc.desc() for pc in pics)
    vp = lambda x: x.ctypes.data_as(C.c_void_p)

    def run_port(k):
        t0 = time.perf_counter()
        rc = orc.svt_oracle_me_picture(C.byref(dc), C.byref(d0), C.byref(d1), C.byref(p), vp(res), None, ranges[k][0], ranges[k][1])
        assert rc == 0
        return time.perf_counter() - t0
    with ThreadPoolExecutor(len(ranges)) as ex:
        port_t = list(ex.map(run_port, range(len(ranges))))
    port_one = run_port(0)
    n_done = sum(s1 - s0 for s0, s1 in ranges)
    same = all(not T.me_results_equal(ref_out[k][1][s0:s1], res[s0:s1], 2) for k, (s0, s1) in enumerate(ranges))
    n0 = ranges[0][1] - ranges[0][0]
    return {"kind": "reference", "what": "motion_estimate_sb (Codec/EbMotionEstimation.c:4524) of the reference built from its own sources, C path (-asm 0), gcc -O2",
            "sample": f"{n_done} of the {nsb} superblocks of one {Wd}x{Hd} B picture (temporal layer {LAYER[i - 1]}), {len(ranges)} processes x {per} SBs",
            "workers": len(ranges),
            "sb_per_s": round(n_done / max(t_ for t_, _ in ref_out), 1), "frames_per_s_me_only": round(n_done / max(t_ for t_, _ in ref_out) / nsb, 3),
            "sb_per_s_1_thread": round(n0 / one[0], 1),
            "port_sb_per_s": round(n_done / max(port_t), 1), "port_sb_per_s_1_thread": round(n0 / port_one, 1),
            "port_over_reference_1_thread": round((n0 / port_one) / (n0 / one[0]), 3),
            "results_identical": bool(same)}


def reference_stage_rates(T, B, src_all, pred_pic, kcell_pic, mc_mi_pic, Wd, Hd, level, ncpu):
    """The REFERENCE's own code for the three stages behind mode decision, timed like reference_me: oracle/_ref/ref_tq_binding
    (perform_coding_loop at its encode-pass call sites, Codec/EbEncDecProcess.c:365-587), ref_lf_binding (eb_vp9_build_mask_frame +
    eb_vp9_loop_filter_frame, VPX/vp9_loopfilter.c:1521-1571) and ref_mc_frame (inter_prediction, Codec/EbIntraPrediction.c:49), built in
    the build container from /root/reference (C path, gcc -O2) and travelling prebuilt.  ONE whole picture of the workload per process
    (picture MINIGOP / 4 of GOP 0: its source, the device's prediction of it, its synthesised partition and motion vectors), one
    process alone and `workers` processes at once; the seconds are the stages' own (clock_gettime inside the harnesses, no request I/O)."""
    need = [os.path.join(ROOT, "oracle", "_ref", n) for n in ("ref_tq_binding", "ref_lf_binding", "ref_mc_frame")]
    if not all(os.path.exists(n) for n in need):
        return None
    i = MINIGOP // 4
    ny, nc = Wd * Hd, Wd * Hd // 4
    planes = lambda a: (np.ascontiguousarray(a[:ny].reshape(Hd, Wd)), np.ascontiguousarray(a[ny:ny + nc].reshape(Hd // 2, Wd // 2)),
                        np.ascontiguousarray(a[ny + nc:ny + 2 * nc].reshape(Hd // 2, Wd // 2)))
    src, pred = planes(src_all[i]), planes(pred_pic)
    mi_rows, mi_cols = Hd // 8, Wd // 8
    # the partition as square inter blocks: kind 0 = four 4x4 per 8x8 unit, 1 = 8x8, 2 = 16x16, 3 = 32x32 (transform = block size)
    lf_mi = np.zeros((mi_rows, mi_cols), dtype=B.LF_MODE_INFO_DTYPE)
    lf_mi["sb_type"] = np.array([0, 3, 6, 9], np.uint8)[kcell_pic]
    lf_mi["tx_size"], lf_mi["is_inter"] = kcell_pic, 1
    cells = np.zeros((mi_rows, mi_cols, 6), np.uint8)      # {sb_type, tx_size, skip, ref_frame[0], mode, segment_id}: inter (LAST_FRAME), NEARESTMV
    cells[..., 0], cells[..., 1], cells[..., 3], cells[..., 4] = np.array([3, 3, 6, 9], np.uint8)[kcell_pic], kcell_pic, 1, 13
    a, b = refs_of(i)
    pad = lambda pl, n: np.ascontiguousarray(np.pad(pl, n, mode="edge"))
    mc_case = dict(mi=mc_mi_pic, width=Wd, height=Hd, mi_rows=mi_rows, mi_cols=mi_cols, use_subpel=1, pad=PAD,
                   refs=[tuple(pad(pl, PAD if k == 0 else PAD // 2) for k, pl in enumerate(planes(src_all[j]))) for j in (a, b)])
    workers = max(1, min(ncpu, 16))
    legs = {"tq": lambda n: T.ref_coding_loop_call_sites(src, pred, lf_mi, Q_INDEX, timing=n),
            "lf": lambda n: T.ref_lf_call_site(src[0], src[1], src[2], cells, level, timing=n),
            "mc": lambda n: T.ref_mc_frame(mc_case, timing=n)}
    what = {"tq": "perform_coding_loop at the encode pass's call sites (Codec/EbEncDecProcess.c:365-587, 3830-3940): residual, forward transform, quantiser, inverse, reconstruction",
            "lf": "eb_vp9_build_mask_frame + eb_vp9_loop_filter_frame (VPX/vp9_loopfilter.c:1521-1571) on a real VP9_COMMON",
            "mc": "inter_prediction (Codec/EbIntraPrediction.c:49 -> VPX/vp9_reconinter.c) for every block and plane"}
    out = {}
    for name, run in legs.items():
        try:
            one = run(1)[0]
            many = run(workers)
        except Exception as e:   # a harness that cannot run this configuration is reported, not fatal
            out[f"reference_{name}"] = {"error": str(e)[-160:]}
            continue
        out[f"reference_{name}"] = {"kind": "reference", "what": what[name] + "; the reference built from its own sources, C path (-asm 0), gcc -O2",
                                    "sample": f"one whole {Wd}x{Hd} picture of the workload per process, {workers} processes at once", "workers": workers,
                                    "seconds_per_picture_1_thread": round(one, 4), "seconds_per_picture_all_busy": round(max(many), 4),
                                    "frames_per_s_stage_only": round(workers / max(many), 2)}
    return out


class Geometry:
    """Layouts in HBM.  Source / prediction pictures: three tight planes one after the other (Y, Cb, Cr).  Reconstruction =
    reference pictures: three padded planes one after the other (Y with PAD samples of border, Cb and Cr with PAD / 2), as the
    reference allocates its reference pictures (Codec/EbEncHandle.c:968-971)."""

    def __init__(self, w, h):
        self.w, self.h = w, h
        self.pic_bytes = w * h * 3 // 2
        self.pw, self.ph = w + 2 * PAD, h + 2 * PAD
        self.cpw, self.cph = w // 2 + PAD, h // 2 + PAD
        self.u_base = self.pw * self.ph
        self.v_base = self.u_base + self.cpw * self.cph
        self.rec_bytes = (self.v_base + self.cpw * self.cph + 63) // 64 * 64
        self.y0 = PAD * self.pw + PAD                               # offsets of sample (0,0) of each plane inside a reference picture
        self.u0 = self.u_base + (PAD // 2) * self.cpw + PAD // 2
        self.v0 = self.v_base + (PAD // 2) * self.cpw + PAD // 2
        self.n_sb = ((w + 63) // 64) * ((h + 63) // 64)
        self.coeffs = self.n_sb * 6144                              # SVT_SB_COEFFS per SB: the driver's position-addressed layout
        self.eob_entries = (w // 4) * (h // 4) * 3 // 2


RING = 8   # mini-GOPs whose reference pictures are alive at a time (the diagonal schedule reaches back LEVELS mini-GOPs); a multiple of the
           # key-frame period in mini-GOPs, so that "first mini-GOP of a closed GOP" is a property of the ring slot
INTRA_PERIOD = 64   # pictures per closed GOP behind its key frame (-intra-period 64 at 60 fps: one key frame per 64 inter pictures)


INTRA_WGS_FROM_CALLER = "SVT_HIP_INTRA_WGS" in os.environ


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--preset", choices=sorted(PRESETS), default=os.environ.get("SVT_BENCH_PRESET", "c3"),
                    help="BASELINE.json configuration: c2 = 1920x1080 enc-mode 8 tune 1, c3 = 3840x2160 enc-mode 8 tune 1 (the one `metric` is quoted on), "
                         "c5 = 3840x2160 enc-mode 3 tune 0 (64x64 search area, SSD fractional search, 3 hierarchical levels)")
    ap.add_argument("--width", type=int, default=0, help="override the preset's picture width (profiling aid)")
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--gops", type=int, default=int(os.environ.get("SVT_BENCH_GOPS", "4")), help="closed GOPs in flight per GPU (value); single_gop_value always uses 1")
    ap.add_argument("--groups", type=int, default=int(os.environ.get("SVT_BENCH_GROUPS", "2")), help="GOP groups = EncDec streams")
    ap.add_argument("--schedule", choices=("diagonal", "waves"), default=os.environ.get("SVT_BENCH_SCHEDULE", "diagonal"),
                    help="EncDec-side schedule of `value`: diagonal = per step, temporal layer l of the mini-GOP l steps back (batches of mutually "
                         "independent pictures per GOP group); waves = the layers of the newest mini-GOP one after the other")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-single", action="store_true", help="skip the G = 1 runs (single_stream_value, single_gop_value)")
    ap.add_argument("--no-extras", action="store_true", help="skip value_reference_flags, the ME-alone pass and the public-API leg (profiling aid)")
    ap.add_argument("--handoff", action="store_true", help="N > 1 only: split-GOP mode -- every step each rank also hands the padded base-layer reconstruction of its "
                    "first GOP to the next rank (RCCL send / recv over xGMI), the one exchange step of the path; closed GOPs (the default) need none")
    ap.add_argument("--no-key-frames", action="store_true", help="profiling aid: inter pictures only (the round-4 step); `value_inter_only` is measured either way")
    ap.add_argument("--stages", default=",".join(STAGES), help="profiling aid: pa, me and / or the EncDec chain (any of mc..pad runs the whole chain)")
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
        # SVT_BENCH_BACKEND=gloo: dry run of the multi-rank control flow on a box with fewer GPUs than ranks (several ranks then share a
        # device, which RCCL refuses); the driver's runs use the default
        dist.init_process_group(os.environ.get("SVT_BENCH_BACKEND", "nccl"))

    import importlib.util
    import svt_testlib as T
    _sp = importlib.util.spec_from_file_location("gop_shard", os.path.join(ROOT, "svt-vp9_amd", "gop_shard.py"))
    GS = importlib.util.module_from_spec(_sp)
    _sp.loader.exec_module(GS)
    B = T.B
    lib = B.load()
    dev = torch.device("cuda", local_rank)
    Wd, Hd, enc_mode, tune = PRESETS[args.preset]
    Wd, Hd = args.width or Wd, args.height or Hd
    set_structure(4 if tune != 0 else 3)
    n_layers = LEVELS + 1
    G = max(1, args.gops)
    n_groups = max(1, min(args.groups, G))
    # The public-API leg (and the intra-pass timing) are separate PROCESSES on the same GPU: they run first, before this process owns a single
    # stream -- with its ~dozen hardware queues alive beside the caller's (torch keeps a stream pool for the life of the process) the 40 ms runs
    # came out 10-25 % lower and noisy (2 500-3 200 against 3 150-3 190 alone, tools/r06_api.py).
    early_legs = {}
    if not args.no_extras and rank == 0 and world == 1 and ((Wd, Hd) == (W4K, H4K) or os.environ.get("SVT_BENCH_API_PATH")):
        api = api_path_rate(T.gen_clip(Wd, Hd, MINIGOP + 1, seed=GS.gop_seed(11, 0)), Wd, Hd, enc_mode, tune)   # (GOP 0's pictures, as frames0 below)
        if api is not None:
            early_legs["api_path"] = api
        ip = intra_pass_times()
        if ip is not None:
            early_legs["intra_pass"] = ip
    t_setup0 = time.perf_counter()

    # streams: ME (+ a second ME stream: ME of different pictures is independent, the tail of one launch is filled by the other),
    # one per GOP group for the EncDec side (the driver's chain, in program order = in dependency order), one for picture
    # analysis.  ME fills every CU by itself: the EncDec streams get the higher priority, so that when an ME workgroup retires a
    # waiting transform workgroup moves in first.
    prio = [int(x) for x in os.environ.get("SVT_BENCH_PRIO", "0,-1").split(",")]
    n_me_streams = max(1, int(os.environ.get("SVT_BENCH_ME_STREAMS", "2")))
    ctxs = []

    def new_ctx(priority):
        st_ = torch.cuda.Stream(device=local_rank, priority=priority)
        c_ = C.c_void_p()
        B.check(lib.svt_hip_ctx_create_on_stream(C.byref(c_), local_rank, C.c_void_p(st_.cuda_stream)))
        ctxs.append(c_)
        return st_, c_

    me_pairs = [new_ctx(prio[0]) for _ in range(n_me_streams)]
    me_streams, me_ctxs = [p_[0] for p_ in me_pairs], [p_[1] for p_ in me_pairs]
    grp_pairs = [new_ctx(prio[1]) for _ in range(n_groups)]
    pa_stream, ctx_pa = new_ctx(prio[0])
    # the key frames' encode pass: beside the inter batches of its GOP group (SVT_BENCH_KEY_STREAMS: how many streams they share)
    key_streams = [new_ctx(prio[1]) for _ in range(max(1, min(n_groups, int(os.environ.get("SVT_BENCH_KEY_STREAMS", str(n_groups))))))]

    # The intra pass of a key frame runs beside the step's other work in the pipelined schedules: 128 workgroups instead of one per CU (the
    # library's default, the lowest latency for a key frame alone: 6.5 vs 7.9 ms) leave half of the CUs with all five motion-estimation
    # workgroups resident -- `value` 5 900 -> 6 030 (tools/bench_intra_wgs.sh, profiles/r05_lf_launch_shape.txt; 64: the pass takes longer
    # than a step).  One GOP at a time (single_*_value) the key frame is on the critical path: the default there.  A deployment knob of the
    # library (svt_hip_ctx_set_intra_workgroups; SVT_HIP_INTRA_WGS in the environment overrides both).
    def key_workgroups(n):
        if not INTRA_WGS_FROM_CALLER:
            for _, c_ in key_streams:
                B.check(lib.svt_hip_ctx_set_intra_workgroups(c_, n))
    single_pairs = [new_ctx(prio[1]) for _ in range(max(1, int(os.environ.get("SVT_BENCH_SINGLE_STREAMS", "2"))))]

    geo = Geometry(Wd, Hd)
    nsbx, nsb = (Wd + 63) // 64, geo.n_sb
    mi_rows, mi_cols = Hd // 8, Wd // 8
    sb_rows, sb_cols = (mi_rows + 7) // 8, (mi_cols + 7) // 8
    pic_bytes = geo.pic_bytes
    keep = []  # keeps device tensors alive

    def to_dev(a):
        t = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        keep.append(t)
        return t

    def dev_zeros(shape, dtype):
        t = torch.zeros(shape, dtype=dtype, device=dev)
        keep.append(t)
        return t

    # ---- synthetic input, resident in HBM: G GOP segments of MINIGOP + 1 pictures (index 0 = the previous mini-GOP's base picture) ----
    d_src = dev_zeros((G, MINIGOP + 1, pic_bytes), torch.uint8)
    frames0, src0 = None, None
    for g in range(G):
        frames = T.gen_clip(Wd, Hd, MINIGOP + 1, seed=GS.gop_seed(11, rank * G + g))   # rank r encodes its own GOP segments
        src_all = np.zeros((MINIGOP + 1, pic_bytes), np.uint8)
        for i, y in enumerate(frames):
            src_all[i, :Wd * Hd] = y.ravel()
            src_all[i, Wd * Hd:Wd * Hd * 5 // 4] = (y[::2, ::2] // 2 + 32).ravel()
            src_all[i, Wd * Hd * 5 // 4:] = (255 - y[::2, ::2] // 2 - y[1::2, 1::2] // 4).ravel()
        d_src[g].copy_(torch.from_numpy(src_all))
        if g == 0:
            frames0, src0 = frames, src_all   # the CPU baseline and the public-API leg work on GOP 0
    src_ptr = lambda g, i: d_src.data_ptr() + (g * (MINIGOP + 1) + i) * pic_bytes

    def tight_desc(d, base):
        d.y, d.u, d.v = base, base + Wd * Hd, base + Wd * Hd * 5 // 4
        d.y_stride, d.uv_stride, d.width, d.height = Wd, Wd // 2, Wd, Hd

    # ---- stage "pa": the three ME planes of every picture from its luma (EbPaReferenceObject planes: padding 68 / 32 / 16) ----
    me_params = lambda layer: B.me_params_preset(Wd, Hd, enc_mode, tune, 2, layer, LEVELS)
    p_probe = me_params(1)
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
    pa_sets = [[[pa_alloc() for _ in range(MINIGOP + 1)] for _ in range(G)] for _ in range(2)]

    def pa_call(ctx_, gops, idx, s=0):
        items = [(g, i) for g in gops for i in idx]
        n = len(items)
        lum = (C.c_void_p * n)(*[src_ptr(g, i) for g, i in items])
        strides = (C.c_int32 * n)(*[Wd] * n)
        out = (B.PaPicture * n)(*[pa_sets[s][g][i] for g, i in items])
        B.check(lib.svt_hip_pa_prepare_batch_device(ctx_, n, lum, strides, out, 1 if l1_on else 0))

    all_gops = list(range(G))
    for s_ in range(2):
        pa_call(me_ctxs[0], all_gops, [0], s_)   # the previous mini-GOP's base picture: analysed when that mini-GOP was
    B.check(lib.svt_hip_ctx_synchronize(me_ctxs[0]))
    pa_idx = list(range(1, MINIGOP + 1))

    # ---- stage "me": one batched launch per temporal layer over the GOPs of the pipeline ----
    results = [[dev_zeros((nsb, 85 * 10), torch.int32) for _ in range(MINIGOP + 1)] for _ in range(G)]

    # Launch grouping: the temporal layers whose parameter sets may share a launch (svt_hip_me_params_same_launch: the library's own rule,
    # the one the public-API path uses to put a whole mini-GOP into one launch) go out together -- fewer, larger launches have fewer
    # tails (a launch's last wave of workgroups leaves CUs idle).  One group is cut into one launch per ME stream.  SVT_BENCH_ME_MERGE=0:
    # one launch per temporal layer (rounds 1-4; the counter passes of tools/profile_round.sh use it: same work, countable launches).
    me_merge = os.environ.get("SVT_BENCH_ME_MERGE", "1") != "0"

    def me_layer_groups():
        if not me_merge:
            return [[layer] for layer in range(n_layers)]
        groups = []
        for layer in range(n_layers):
            p_l = me_params(layer)
            for gset in groups:
                p_0 = me_params(gset[0])
                if lib.svt_hip_me_params_same_launch(C.byref(p_0), C.byref(p_l)):
                    gset.append(layer)
                    break
            else:
                groups.append([layer])
        return groups

    def build_me_launches(gops, one_stream=False):
        sets = []
        groups = me_layer_groups()
        for s in range(2):
            launches = []
            chunks = []
            for gset in groups:
                idx = [i for layer in gset for i in pics_of_layer(layer)]
                items = [(g, i) for g in gops for i in idx]
                n_cut = 1 if (one_stream or not me_merge or len(groups) >= len(me_ctxs)) else min(len(me_ctxs), len(items))
                for k_ in range(n_cut):   # (interleaved: every chunk holds pictures of every layer)
                    chunks.append(items[k_::n_cut])
            for items in chunks:
                n = len(items)
                p = (B.MeParams * n)()
                for k_, (g, i) in enumerate(items):
                    p[k_] = me_params(LAYER[i - 1])
                    p[k_].same_ref_poc = 1 if LAYER[i - 1] == 0 else 0
                cur = (B.PaPicture * n)(*[pa_sets[s][g][i] for g, i in items])
                r0 = (B.PaPicture * n)(*[pa_sets[s][g][refs_of(i)[0]] for g, i in items])
                r1 = (B.PaPicture * n)(*[pa_sets[s][g][refs_of(i)[1]] for g, i in items])
                res = (C.c_void_p * n)(*[results[g][i].data_ptr() for g, i in items])
                launches.append((n, cur, r0, r1, p, res))
            sets.append(launches)
        # launches -> ME streams: largest first onto the least loaded stream
        slot, load = [0] * len(sets[0]), [0] * len(me_ctxs)
        if not one_stream:
            for li in sorted(range(len(sets[0])), key=lambda j: -sets[0][j][0]):
                k_ = load.index(min(load))
                slot[li] = k_
                load[k_] += sets[0][li][0]
        return sets, slot

    # event pools are created before the timed region: creating an event costs the host tens of microseconds and, inside the
    # launch path, the overlap between streams (measured in round 2); recording an existing event costs nothing measurable
    class EventPool:
        def __init__(self, n):
            self.ev = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
            for e_ in self.ev:
                e_.record(me_streams[0])   # an event object is created on its first record

        def take(self):
            return self.ev.pop()

    def run_me(P, s, S=None, pool=None):
        for st_ in me_streams[1:]:
            st_.wait_stream(me_streams[0])
        for li, (n, cur, r0, r1, p, res) in enumerate(P["me_sets"][s]):
            k_ = P["me_slot"][li]
            if pool is not None:
                e0 = pool.take()
                e0.record(me_streams[k_])
            B.check(lib.svt_hip_me_batch_layers_device(me_ctxs[k_], n, cur, r0, r1, p, res, None))
            if pool is not None:
                e1 = pool.take()
                e1.record(me_streams[k_])
                S["ev"].append(("me", k_, e0, e1))
        for st_ in me_streams[1:]:
            me_streams[0].wait_stream(st_)

    # ---- first pass (setup, untimed): PA + ME of every GOP, results to the host = the input of the synthesised mode decision ----
    P_all = {"gops": all_gops}
    P_all["me_sets"], P_all["me_slot"] = build_me_launches(all_gops)
    pa_call(me_ctxs[0], all_gops, pa_idx)
    with torch.cuda.stream(me_streams[0]):
        run_me(P_all, 0)
    for c_ in ctxs:
        B.check(lib.svt_hip_ctx_synchronize(c_))
    torch.cuda.synchronize()

    ac_q = lib.svt_hip_vp9_ac_step(Q_INDEX)
    level = lib.svt_hip_lf_level_from_q(ac_q, 0)
    thr = B.LfThresh()
    lib.svt_hip_lf_thresh_init(C.byref(thr), 0)
    d_mc = [[None] * (MINIGOP + 1) for _ in range(G)]
    d_lf = [[None] * (MINIGOP + 1) for _ in range(G)]
    kcell = [[None] * (MINIGOP + 1) for _ in range(G)]
    mc_host0, lf_host0 = {}, {}
    inter_units = comp_units = mi_units = 0
    blocks_by_size = np.zeros(4, np.int64)
    for g in range(G):
        rng = np.random.default_rng(5 + rank * G + g)
        for i in range(1, MINIGOP + 1):
            res = results[g][i].cpu().numpy().view(B.ME_RESULT_DTYPE).reshape(nsb, 85)
            kinds = partition_kinds(rng, Wd, Hd)
            mi, k_cell = build_mode_info(B, res, kinds, mi_rows, mi_cols, nsbx)
            lmi = build_lf_mode_info(B, k_cell, mi_rows, mi_cols, level)
            d_mc[g][i], d_lf[g][i], kcell[g][i] = to_dev(mi.view(np.uint8)), to_dev(lmi.view(np.uint8)), k_cell
            inter, comp = mi["ref_list"][..., 0] >= 0, mi["ref_list"][..., 1] >= 0
            inter_units += int(inter.sum())
            comp_units += int((inter & comp).sum())
            mi_units += inter.size
            n0, n1, n2, n3 = int((k_cell == 0).sum()), int((k_cell == 1).sum()), int((k_cell == 2).sum()) // 4, int((k_cell == 3).sum()) // 16
            # transform blocks: 8x8 block / 4x4: 4 luma + 2 chroma 4x4; 8x8 / 8x8: 1 luma 8x8 + 2 chroma 4x4; 16x16: 1 + 2 chroma 8x8; 32x32: 1 + 2 chroma 16x16
            blocks_by_size += np.array([6 * n0 + 2 * n1, n1 + 2 * n2, n2 + 2 * n3, n3])
            if g == 0:
                mc_host0[i], lf_host0[i] = mi, lmi

    # ---- EncDec-side arenas: prediction pictures, reference pictures (= reconstruction buffers), coefficients, per-picture maps ----
    # Reference pictures live in a ring of RING mini-GOPs per GOP: picture i (1..MINIGOP) of mini-GOP m is rec[m % RING][g][i - 1]; its
    # "picture 0" -- the base picture of the mini-GOP before -- is rec[(m - 1) % RING][g][MINIGOP - 1].  Nothing is ever copied: a
    # picture is reconstructed, deblocked and padded in the buffer later pictures predict from.
    d_pred = dev_zeros((G, MINIGOP, pic_bytes), torch.uint8)
    d_rec = dev_zeros((RING, G, MINIGOP, geo.rec_bytes), torch.uint8)
    d_q = dev_zeros(G * MINIGOP * geo.coeffs, torch.int16)
    d_emap = dev_zeros((G, MINIGOP, geo.eob_entries), torch.int16)
    d_lfm = dev_zeros((G, MINIGOP, nsb * B.LF_MASK_DTYPE.itemsize), torch.uint8)
    d_nz = dev_zeros((G, MINIGOP, mi_rows * mi_cols), torch.uint8)
    assert max(d_src.numel(), d_pred.numel()) < 2 ** 32 and d_q.numel() < 2 ** 32, "32-bit block offsets: fewer GOPs in flight"
    rec_ptr = lambda slot, g, i: d_rec.data_ptr() + ((slot % RING) * G * MINIGOP + g * MINIGOP + i - 1) * geo.rec_bytes
    pred_ptr = lambda g, i: d_pred.data_ptr() + (g * MINIGOP + i - 1) * pic_bytes

    def yuv_desc(d, slot, g, i):
        base = rec_ptr(slot, g, i)
        d.y, d.u, d.v = base + geo.y0, base + geo.u0, base + geo.v0
        d.y_stride, d.uv_stride, d.width, d.height = geo.pw, geo.cpw, Wd, Hd

    # Key frames.  A closed GOP is a key frame + INTRA_PERIOD inter pictures = KEYP mini-GOPs; GOP stream g starts a new closed GOP in
    # the ring slots with slot % KEYP == g % KEYP (staggered: with G = KEYP one key frame per step).  The first mini-GOP of a closed
    # GOP predicts from the key frame's reconstruction (its "picture 0"), not from the base picture of the mini-GOP before.  Two key
    # buffers per stream alternate: the deepest layer of that first mini-GOP still reads its key frame LEVELS steps later.
    KEYP = max(1, INTRA_PERIOD // MINIGOP)
    KEY_LEAD = int(os.environ.get("SVT_BENCH_KEY_LEAD", "2"))   # steps between a key frame's encode pass and the first batch that predicts from it
    use_keys = [not args.no_key_frames]
    d_key = dev_zeros((G, 2, geo.rec_bytes), torch.uint8)
    key_first = lambda slot, g: use_keys[0] and (slot % RING) % KEYP == g % KEYP
    key_buf = lambda slot: ((slot % RING) // KEYP) & 1

    def key_desc(d, g, k):
        base = d_key.data_ptr() + (g * 2 + k) * geo.rec_bytes
        d.y, d.u, d.v = base + geo.y0, base + geo.u0, base + geo.v0
        d.y_stride, d.uv_stride, d.width, d.height = geo.pw, geo.cpw, Wd, Hd

    def ref_desc(d, slot, g, j):
        """reference picture j (0..MINIGOP) of the mini-GOP in ring slot `slot`"""
        if j == 0 and key_first(slot, g):
            key_desc(d, g, key_buf(slot))
        elif j == 0:
            yuv_desc(d, slot - 1, g, MINIGOP)
        else:
            yuv_desc(d, slot, g, j)

    # the base picture of the mini-GOP before the first one (ring slot -1): its source, padded -- from then on every reference is
    # a reconstruction
    for g in range(G):
        s0 = d_src[g, 0]
        y, u, v = s0[:Wd * Hd].view(Hd, Wd), s0[Wd * Hd:Wd * Hd * 5 // 4].view(Hd // 2, Wd // 2), s0[Wd * Hd * 5 // 4:].view(Hd // 2, Wd // 2)
        r = d_rec[RING - 1, g, MINIGOP - 1]
        r[:geo.u_base].view(geo.ph, geo.pw)[PAD:PAD + Hd, PAD:PAD + Wd] = y
        r[geo.u_base:geo.v_base].view(geo.cph, geo.cpw)[PAD // 2:PAD // 2 + Hd // 2, PAD // 2:PAD // 2 + Wd // 2] = u
        r[geo.v_base:geo.v_base + geo.cph * geo.cpw].view(geo.cph, geo.cpw)[PAD // 2:PAD // 2 + Hd // 2, PAD // 2:PAD // 2 + Wd // 2] = v
    dsc0 = (B.YuvPlanes * G)()
    for g in range(G):
        yuv_desc(dsc0[g], -1, g, MINIGOP)
    torch.cuda.synchronize()
    B.check(lib.svt_hip_ref_pad_batch_device(me_ctxs[0], G, dsc0, PAD, PAD))
    B.check(lib.svt_hip_ctx_synchronize(me_ctxs[0]))

    def flags_for(layer, recon_file):
        c_ = B.EncdecFlagsConfig(enc_mode=enc_mode, tune=tune, temporal_layer_index=layer, is_used_as_reference=int(layer < LEVELS), recon_file=recon_file, loop_filter=1)
        f_ = B.EncdecFlags()
        B.check(lib.svt_hip_encdec_flags_derive(C.byref(c_), C.byref(f_)))
        return f_

    def build_batches(items, recon_file):
        """driver calls of one set of mutually independent pictures: items = (g, i, back) -- picture i of GOP g of the mini-GOP `back`
        mini-GOPs before the newest one of the step.  Pictures with the same stage flags share a call, at most ED_BATCH per call; one
        descriptor array per ring phase."""
        classes = {}
        for it in items:   # pictures that differ only in "is padded" ride in one call (svt_encdec_picture.no_pad)
            f_ = flags_for(LAYER[it[1] - 1], recon_file)
            c_ = classes.setdefault((f_.do_recon, f_.apply_loop_filter), [f_, [], []])
            c_[1].append(it)
            c_[2].append(0 if f_.pad_reference else 1)
            if f_.pad_reference:
                c_[0] = f_
        out = []
        for key in sorted(classes, reverse=True):
            f_, its, nopad = classes[key]
            for b0 in range(0, len(its), ED_BATCH):
                chunk, chunk_nopad = its[b0:b0 + ED_BATCH], nopad[b0:b0 + ED_BATCH]
                n = len(chunk)
                per_phase = []
                for ph in range(RING):
                    arr = (B.EncdecPicture * n)()
                    for k, (g, i, back) in enumerate(chunk):
                        slot, p = ph - back, arr[k]
                        p.d_mc_mi, p.d_lf_mi = d_mc[g][i].data_ptr(), d_lf[g][i].data_ptr()
                        tight_desc(p.src, src_ptr(g, i))
                        tight_desc(p.pred, pred_ptr(g, i))
                        yuv_desc(p.recon, slot, g, i)
                        for l in range(2):
                            ref_desc(p.ref[l], slot, g, refs_of(i)[l])
                        co = (g * MINIGOP + i - 1) * geo.coeffs * 2
                        p.d_qcoeff, p.d_dqcoeff = d_q.data_ptr() + co, None   # the encode pass does not need dqcoeff in memory (svt_encdec_picture)
                        p.d_eob_map, p.d_lfm, p.d_nz = d_emap[g, i - 1].data_ptr(), d_lfm[g, i - 1].data_ptr(), d_nz[g, i - 1].data_ptr()
                        p.use_subpel, p.no_pad = 1, chunk_nopad[k]
                    per_phase.append(arr)
                out.append({"n": n, "flags": f_, "pics": per_phase, "items": chunk})
        return out

    HOOK = C.CFUNCTYPE(None, C.c_void_p, C.c_int32)

    # ---- the encode pass of a key frame: stand-in decision (16x16 DC: the reference's intra mode decision is out of scope) once in
    # set-up, svt_hip_encdec_intra_device (prediction + transform wavefront, deblocking, border) inside the step ----
    level_key = lib.svt_hip_lf_level_from_q(ac_q, 1)
    d_key_mi = dev_zeros((mi_rows * mi_cols, B.LF_MODE_INFO_DTYPE.itemsize), torch.uint8)
    B.check(lib.svt_hip_md_intra_default_device(me_ctxs[0], Wd, Hd, level_key, C.c_void_p(d_key_mi.data_ptr()), mi_cols))
    B.check(lib.svt_hip_ctx_synchronize(me_ctxs[0]))
    fl_key_cfg = B.EncdecFlagsConfig(enc_mode=enc_mode, tune=tune, temporal_layer_index=0, is_used_as_reference=1, recon_file=1, loop_filter=1)
    fl_key = B.EncdecFlags()
    B.check(lib.svt_hip_encdec_flags_derive(C.byref(fl_key_cfg), C.byref(fl_key)))

    def key_resources(ctx_):
        """what one stream of key frames needs: a driver workspace and the per-picture outputs (coefficients, eob map, masks) -- nothing
        downstream consumes them here, so one set per stream"""
        work = C.c_void_p()
        B.check(lib.svt_hip_encdec_work_create(ctx_, 1, Wd, Hd, C.byref(work)))
        return {"work": work, "q": dev_zeros(geo.coeffs, torch.int16), "emap": dev_zeros(geo.eob_entries, torch.int16),
                "lfm": dev_zeros(nsb * B.LF_MASK_DTYPE.itemsize, torch.uint8), "nz": dev_zeros(mi_rows * mi_cols, torch.uint8)}

    def key_picture(res, g, k):
        p = B.EncdecPicture()
        p.d_lf_mi = d_key_mi.data_ptr()
        tight_desc(p.src, src_ptr(g, 0))
        key_desc(p.recon, g, k)
        p.d_qcoeff, p.d_dqcoeff, p.d_eob_map, p.d_lfm, p.d_nz = res["q"].data_ptr(), None, res["emap"].data_ptr(), res["lfm"].data_ptr(), res["nz"].data_ptr()
        return p

    def run_key(ctx_, res, pic):
        B.check(lib.svt_hip_encdec_intra_device(ctx_, res["work"], C.byref(pic), Wd, Hd, mi_cols, Q_INDEX, C.byref(fl_key), C.byref(thr), PAD, PAD))

    def build_pipeline(gops, pairs, split="gop", recon_file=1, keys=True):
        """a pipeline = a set of GOPs in flight whose pictures are split into groups with one EncDec stream each: whole GOPs per
        group ("gop"), or -- any partition of a batch of mutually independent pictures is valid -- the pictures of every GOP dealt
        round-robin ("picture": what one stream of mini-GOPs uses to overlap the deblocking of one half of a batch with the transform
        stage of the other).  Per group: the wave batches (the group's pictures of one layer of the newest mini-GOP) and the diagonal
        batches (layer l of the mini-GOP l steps back), and the driver workspace of its context."""
        ng = len(pairs)
        if split == "gop":
            members = [[(g, i) for g in gops[k::ng] for i in range(1, MINIGOP + 1)] for k in range(ng)]
        else:
            members = [[(g, i) for g in gops for i in range(1, MINIGOP + 1) if i % ng == k] for k in range(ng)]
        keys = keys and not args.no_key_frames
        use_keys[0] = keys          # read by ref_desc while the descriptors below are built
        P = {"gops": gops, "groups": [], "keys": keys, "n_keys": 0, "key_ready": {}, "key_owner": {}}
        # with the pictures of a GOP spread over several streams, a batch's references may have been finished on another stream:
        # every group then waits for the events all groups recorded behind their previous batch (events from a pool made here)
        P["cross_sync"] = split == "picture" and ng > 1
        P["sync_pool"] = [torch.cuda.Event() for _ in range(64 * ng)] if P["cross_sync"] else []
        for e_ in P["sync_pool"]:
            e_.record(me_streams[0])
        P["sync_pos"], P["sync_last"] = 0, []
        P["me_sets"], P["me_slot"] = build_me_launches(gops)
        for k_, (pics_, (st_, ctx_)) in enumerate(zip(members, pairs)):
            work = C.c_void_p()
            B.check(lib.svt_hip_encdec_work_create(ctx_, ED_BATCH, Wd, Hd, C.byref(work)))
            grp_gops = sorted({g for g, _ in pics_})
            ist_, ictx_ = key_streams[k_ % len(key_streams)]
            kres = key_resources(ictx_) if keys else None
            grp = {"pics": pics_, "stream": st_, "ctx": ctx_, "work": work, "index": k_, "gops": grp_gops, "istream": ist_, "ictx": ictx_, "kres": kres,
                   "kpics": {(g, k): key_picture(kres, g, k) for g in grp_gops for k in range(2)} if keys else {},
                   "ev_ring": [torch.cuda.Event() for _ in range(8)], "ev_pos": 0, "step_done": None,
                   "waves": [build_batches([(g, i, 0) for g, i in pics_ if LAYER[i - 1] == layer], recon_file) for layer in range(n_layers)],
                   "diag": build_batches([(g, i, LAYER[i - 1]) for g, i in pics_], recon_file)}
            for e_ in grp["ev_ring"]:
                e_.record(st_)
            for g in grp_gops:
                P["key_owner"].setdefault(g, k_)
            if keys:    # every key buffer holds a real reconstruction before the first step
                with torch.cuda.stream(ist_):
                    for (g, _k), kp in grp["kpics"].items():
                        if P["key_owner"][g] == k_:
                            run_key(ictx_, kres, kp)
            P["groups"].append(grp)
        use_keys[0] = not args.no_key_frames
        return P

    def run_batches(grp, batches, ph):
        for bt in batches:
            B.check(lib.svt_hip_encdec_batch_device(grp["ctx"], grp["work"], bt["n"], bt["pics"][ph], Wd, Hd, mi_cols, Q_INDEX, C.byref(bt["flags"]), C.byref(thr), PAD, PAD))

    P_main = build_pipeline(all_gops, grp_pairs)

    # ---- setup pass of the dependent chain (untimed, mini-GOP 0 = ring phase 0, wave schedule) -- it fills ring slot 0 with real
    # reconstructions and yields the workload statistics ----
    for layer in range(n_layers):
        for grp in P_main["groups"]:
            with torch.cuda.stream(grp["stream"]):
                run_batches(grp, grp["waves"][layer], 0)
    for grp in P_main["groups"]:
        cnt = (C.c_int32 * 8)()
        B.check(lib.svt_hip_encdec_work_status(grp["ctx"], grp["work"], cnt))     # synchronises; raises on a malformed grid
    resid_acc, eob_by_size = [], [[] for _ in range(4)]
    with torch.no_grad():
        for i in range(1, MINIGOP + 1):
            sy, py = d_src[0, i, :Wd * Hd].to(torch.int16), d_pred[0, i - 1, :Wd * Hd].to(torch.int16)
            resid_acc.append((sy - py).abs().to(torch.float32).mean().item())
            em = d_emap[0, i - 1, :(Wd // 4) * (Hd // 4)].cpu().numpy().view(np.uint16).reshape(Hd // 4, Wd // 4)
            k4 = np.kron(kcell[0][i], np.ones((2, 2), np.int64))[:Hd // 4, :Wd // 4]
            y4, x4 = np.meshgrid(np.arange(Hd // 4), np.arange(Wd // 4), indexing="ij")
            for k in range(4):
                n4 = 1 << k
                eob_by_size[k].append(em[(k4 == k) & (y4 % n4 == 0) & (x4 % n4 == 0)].astype(np.int64))
    skip_share = float(np.mean([d_lf[0][i].cpu().numpy().view(B.LF_MODE_INFO_DTYPE)["skip"].mean() for i in range(1, MINIGOP + 1)]))
    workload_stats = {"mean_abs_luma_residual_gop0": round(float(np.mean(resid_acc)), 2),
                      "mean_luma_eob_by_tx_size": [round(float(np.concatenate(eob_by_size[ts]).mean()), 1) if sum(len(a) for a in eob_by_size[ts]) else None for ts in range(4)],
                      "transform_blocks_by_tx_size_per_step": [int(v) for v in blocks_by_size],
                      # the synthesised partition: share of the step's transform blocks / of its samples per transform size (4x4, 8x8, 16x16, 32x32)
                      "tx_size_share_of_blocks": [round(float(v) / max(1, int(blocks_by_size.sum())), 3) for v in blocks_by_size],
                      "tx_size_share_of_samples": [round(float(v) * (16 << (2 * k_)) / max(1.0, float(sum(int(b_) * (16 << (2 * j_)) for j_, b_ in enumerate(blocks_by_size)))), 3)
                                                   for k_, v in enumerate(blocks_by_size)],
                      "skip_block_share_gop0": round(skip_share, 3)}
    step_no = [1]   # the setup pass was step 0 (mini-GOP 0 complete in ring slot 0)
    extras = not args.no_extras and rank == 0
    P_single = None if (args.no_single or rank != 0) else build_pipeline([0], single_pairs, split="picture")
    P_ref = build_pipeline(all_gops, grp_pairs, recon_file=0) if extras else None
    P_inter = build_pipeline(all_gops, grp_pairs, keys=False) if (extras and P_main["keys"]) else None
    setup_s = time.perf_counter() - t_setup0

    # split-GOP hand-off (optional, N > 1): the padded base-layer reconstruction of this rank's first GOP, as the deblocking +
    # padding of this step left it (a whole reference picture: luma + chroma with 80 / 40 samples of border)
    handoff = args.handoff and world > 1
    if handoff:
        ho_recv = dev_zeros(geo.rec_bytes, torch.uint8)
        ho_stream = torch.cuda.Stream(device=local_rank)

    def run_handoff(P, ph):
        for grp0 in P["groups"]:
            ho_stream.wait_stream(grp0["stream"])   # ordered after this step's deblocking + padding of the base picture
        with torch.cuda.stream(ho_stream):
            ops = [dist.P2POp(dist.isend, d_rec[ph % RING, P["gops"][0], MINIGOP - 1], (rank + 1) % world), dist.P2POp(dist.irecv, ho_recv, (rank - 1) % world)]
            for w_ in dist.batch_isend_irecv(ops):
                w_.wait()

    stages = set(args.stages.split(","))
    run_encdec = bool(stages & set(ED_STAGE_NAMES))

    def make_state():
        return {"pa_done": [None, None], "me_done": [None, None], "step": 0, "ev": []}

    def step(P, S, schedule, pool=None):
        buf = S["step"] & 1
        S["step"] += 1
        ph = step_no[0] % RING          # ring slot of the newest mini-GOP of this step
        step_no[0] += 1
        # ME side (one mini-GOP ahead of the EncDec side, as the reference's ME threads are): PA writes the planes ME reads and
        # picture analysis of the NEXT mini-GOP runs beside it on its own stream
        if S["pa_done"][buf] is not None:
            me_streams[0].wait_event(S["pa_done"][buf])
        if "me" in stages:
            run_me(P, buf, S, pool)
            S["me_done"][buf] = torch.cuda.Event()
            S["me_done"][buf].record(me_streams[0])
        if "pa" in stages:
            if S["me_done"][1 - buf] is not None:
                pa_stream.wait_event(S["me_done"][1 - buf])   # ME of the previous step read the set written now
            if pool is not None:
                e0 = pool.take()
                e0.record(pa_stream)
            pa_call(ctx_pa, P["gops"], pa_idx, 1 - buf)
            if pool is not None:
                e1 = pool.take()
                e1.record(pa_stream)
                S["ev"].append(("pa", 0, e0, e1))
            S["pa_done"][1 - buf] = torch.cuda.Event()
            S["pa_done"][1 - buf].record(pa_stream)
        if not run_encdec:
            return

        # EncDec side, every stage on the group's stream in program order = dependency order
        def barrier_groups():
            """cross-stream dependency of a picture-split pipeline: the next batch of every group starts after the previous batch of all"""
            if P["cross_sync"]:
                for grp in P["groups"]:
                    for e_ in P["sync_last"]:
                        grp["stream"].wait_event(e_)

        def mark_groups():
            if not P["cross_sync"]:
                return
            P["sync_last"] = []
            for grp in P["groups"]:
                e_ = P["sync_pool"][P["sync_pos"] % len(P["sync_pool"])]
                P["sync_pos"] += 1
                e_.record(grp["stream"])
                P["sync_last"].append(e_)

        def hooked(grp, batches):
            """run the group's driver calls; with a pool, a stage hook records an event at every stage boundary of the driver on the
            group's stream, so that the chain's time is attributed to its stages"""
            if pool is None:
                return run_batches(grp, batches, ph)
            marks = []

            def hook(_user, stage):
                e_ = pool.take()
                e_.record(grp["stream"])
                marks.append((stage, e_))
            cb = HOOK(hook)
            lib.svt_hip_encdec_work_set_stage_hook(grp["work"], cb, None)
            run_batches(grp, batches, ph)
            lib.svt_hip_encdec_work_set_stage_hook(grp["work"], None, None)
            for (s0, e0), (s1, e1) in zip(marks, marks[1:]):
                if s0 < len(ED_STAGE_NAMES):
                    S["ev"].append((ED_STAGE_NAMES[s0], 100 + grp["index"], e0, e1))

        def grp_event(grp, stream):
            e_ = grp["ev_ring"][grp["ev_pos"] % len(grp["ev_ring"])]
            grp["ev_pos"] += 1
            e_.record(stream)
            return e_

        if P["keys"]:
            # a closed GOP whose first mini-GOP is the NEXT step's newest one: its key frame is coded now, on the group's key stream,
            # beside this step's inter batches (the source is there: the ME side already runs a mini-GOP ahead) -- the picture-level
            # overlap the reference's EncDec processes have at a closed-GOP boundary.  The buffer's previous readers are behind the
            # group's last batch; the next step's batch waits for the key frame.
            for grp in P["groups"]:     # (a picture-split pipeline shares its GOPs between the groups: the first group codes the key frames)
                for g in grp["gops"]:
                    if g in P["key_ready"] and P["key_ready"][g][1] <= S["step"]:
                        grp["stream"].wait_event(P["key_ready"][g][0])
            P["key_ready"] = {g: v for g, v in P["key_ready"].items() if v[1] > S["step"]}
            for grp in P["groups"]:
                todo = [g for g in grp["gops"] if ((ph + KEY_LEAD) % RING) % KEYP == g % KEYP and P["key_owner"][g] == grp["index"]]
                if todo:
                    for other in P["groups"]:
                        if other["step_done"] is not None and set(todo) & set(other["gops"]):
                            grp["istream"].wait_event(other["step_done"])
                    with torch.cuda.stream(grp["istream"]):
                        for g in todo:
                            run_key(grp["ictx"], grp["kres"], grp["kpics"][(g, key_buf(ph + KEY_LEAD))])
                            P["n_keys"] += 1
                    e_ = grp_event(grp, grp["istream"])
                    for g in todo:
                        P["key_ready"][g] = (e_, S["step"] + KEY_LEAD)   # S["step"] already counts this step

        if schedule == "waves":      # the dependent temporal-layer waves of the newest mini-GOP
            for layer in range(n_layers):
                barrier_groups()
                for grp in P["groups"]:
                    with torch.cuda.stream(grp["stream"]):
                        hooked(grp, grp["waves"][layer])
                mark_groups()
        else:                        # diagonal: layer l of the mini-GOP l steps back -- batches of independent pictures
            barrier_groups()
            for grp in P["groups"]:
                with torch.cuda.stream(grp["stream"]):
                    hooked(grp, grp["diag"])
            mark_groups()
        if P["keys"]:
            for grp in P["groups"]:
                grp["step_done"] = grp_event(grp, grp["stream"])
        if handoff:
            run_handoff(P, ph)

    def sync():
        for c_ in ctxs:
            B.check(lib.svt_hip_ctx_synchronize(c_))
        torch.cuda.synchronize()

    def timed_run(P, schedule, steps, warmup, barrier):
        S = make_state()
        if schedule == "diagonal":
            warmup = max(warmup, n_layers)   # the pipeline of n_layers mini-GOPs has to fill
        n_calls = sum(len(grp["diag"]) + sum(len(w_) for w_ in grp["waves"]) for grp in P["groups"])
        pool = EventPool(steps * (2 * len(P["me_sets"][0]) + 2 + 8 * n_calls) + 16)
        for _ in range(warmup):
            step(P, S, schedule)
        sync()
        P["n_keys"] = 0
        if barrier and world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        # host cost of a step = time to enqueue it while nothing holds the enqueue thread back: over the first steps only -- once
        # the 64-slot descriptor rings of the contexts are full the thread waits for the GPU
        n_free = min(steps, 8)
        t_enq = 0.0
        for i_ in range(steps):
            step(P, S, schedule, pool)
            if i_ + 1 == n_free:
                t_enq = time.perf_counter() - t0
        sync()
        if barrier and world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        # per stage: the busiest stream's total (what the stage occupies of the step's wall time) and the sum over all streams
        per_stream = {}
        for name, sid, e0, e1 in S["ev"]:
            per_stream.setdefault(name, {}).setdefault(sid, 0.0)
            per_stream[name][sid] += e0.elapsed_time(e1) / steps
        stage_ms = {s: max(per_stream.get(s, {0: 0.0}).values()) for s in STAGES}
        stage_ms_sum = {s: sum(per_stream.get(s, {0: 0.0}).values()) for s in STAGES}
        me_launches = [(e0.elapsed_time(e1)) for name, sid, e0, e1 in S["ev"] if name == "me"]
        return dt, t_enq / n_free, stage_ms, stage_ms_sum, me_launches

    key_workgroups(int(os.environ.get("SVT_BENCH_KEY_WGS", "128")))
    dt, enq_s, stage_ms, stage_ms_sum, me_launch_list = timed_run(P_main, args.schedule, args.steps, args.warmup, True)
    n_keys_main = P_main["n_keys"]
    dt = GS.reduce_elapsed(dt, dist if world > 1 else None, dev if os.environ.get("SVT_BENCH_BACKEND", "nccl") == "nccl" else None)
    if world > 1:   # nothing below needs the other ranks: they leave in step, rank 0 reports
        sync()
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        return
    single = {}
    if P_single is not None:
        k1 = max(4, args.steps)
        for sched in ("diagonal", "waves"):
            key_workgroups(0)
            dt1, enq1, stage1, _, _ = timed_run(P_single, sched, k1, max(2, args.warmup), False)
            key_workgroups(int(os.environ.get("SVT_BENCH_KEY_WGS", "128")))
            single[sched] = {"frames_per_s": (MINIGOP * k1 + P_single["n_keys"]) / dt1, "ms_per_minigop": dt1 / k1 * 1e3, "steps": k1, "stage_ms": stage1, "enq": enq1}
    ref_flags = None
    me_alone = None
    inter_only = key_alone_ms = None
    if extras:
        if P_inter is not None:
            # the same step without the key frames (round 4's `value`), and a key frame's encode pass alone on the idle GPU: what part of
            # it the schedule hides = 1 - (step with - step without) / (key frames per step x alone)
            k0 = max(4, args.steps // 2)
            dt0, _, _, _, _ = timed_run(P_inter, args.schedule, k0, args.warmup, False)
            inter_only = {"frames_per_s": G * MINIGOP * k0 / dt0, "ms_per_step": dt0 / k0 * 1e3, "steps": k0}
            grp0 = P_main["groups"][0]
            ts_ = []
            for _ in range(4):
                sync()
                t_ = time.perf_counter()
                run_key(grp0["ictx"], grp0["kres"], grp0["kpics"][(grp0["gops"][0], 0)])
                B.check(lib.svt_hip_ctx_synchronize(grp0["ictx"]))
                ts_.append(time.perf_counter() - t_)
            key_alone_ms = min(ts_) * 1e3
        k2 = max(4, args.steps // 2)
        dt2, _, stage2, _, _ = timed_run(P_ref, args.schedule, k2, args.warmup, False)
        fl = {layer: flags_for(layer, 0) for layer in range(n_layers)}
        ref_flags = {"frames_per_s": (G * MINIGOP * k2 + P_ref["n_keys"]) / dt2, "ms_per_minigop": dt2 / k2 / G * 1e3, "steps": k2,
                     "flags_by_layer": {str(layer): {"do_recon": fl[layer].do_recon, "deblock": fl[layer].apply_loop_filter, "pad": fl[layer].pad_reference} for layer in fl}}
        # motion estimation alone: one stream, nothing beside it -- the clean per-launch duration of the dominant kernel
        saved, stages = stages, {"me"}
        run_encdec = False
        P_me = {"gops": all_gops, "groups": [], "cross_sync": False}
        P_me["me_sets"], P_me["me_slot"] = build_me_launches(all_gops, one_stream=True)
        k3 = max(4, args.steps // 2)
        dt3, _, _, _, me3 = timed_run(P_me, "waves", k3, 2, False)
        me_alone = {"ms_per_step": dt3 / k3 * 1e3, "launch_ms_sum_per_step": sum(me3) / k3, "steps": k3}
        stages, run_encdec = saved, bool(saved & set(ED_STAGE_NAMES))

    L = Wd * Hd
    pics_step = G * MINIGOP
    n_launch_step = len(P_me["me_sets"][0]) if me_alone else len(P_main["me_sets"][0])   # the launches behind `me_clean_ms`
    n_blocks_step = int(blocks_by_size.sum())
    stage_bytes = {
        # source luma read + padded / decimated planes written
        "pa": pics_step * int(L + (Wd + 2 * pads[0]) * (Hd + 2 * pads[0]) + (Wd // 4 + 2 * pads[2]) * (Hd // 4 + 2 * pads[2]) +
                              (((Wd // 2 + 2 * pads[1]) * (Hd // 2 + 2 * pads[1])) if l1_on else 0)),
        # per picture: R = the number of DISTINCT reference pictures its lists name (the base-layer picture of a mini-GOP has both lists on
        # the previous base picture: R = 1; every other picture is a B picture between two different references: R = 2)
        "me": G * sum(algorithmic_bytes_me(Wd, Hd, len(set(refs_of(i))), l1_on) for i in range(1, MINIGOP + 1)),
        # per 8x8 unit: 96 bytes (64 luma + 2 x 16 chroma) read per reference and written once, + the 12-byte mode-info record
        "mc": int(96 * (inter_units + comp_units) + 96 * inter_units + 12 * mi_units),
        # the 8-byte grid records read twice (count, emit), a 32-byte descriptor + 4-byte position code written per block
        "lists": int(2 * 8 * mi_units + 36 * n_blocks_step),
        "tq": pics_step * int(7.5 * L),                # src 1.5L + pred 1.5L + qcoeff 3L + recon 1.5L (dqcoeff stays in registers: d_dqcoeff = NULL)
        # position code + eob read, eob map written (cleared first), the grid's skip flags updated
        "skip": int(6 * n_blocks_step + 2 * 2 * geo.eob_entries * pics_step + 2 * 8 * mi_units),
        "lf": pics_step * (3 * L + 160 * nsb),         # recon read + write (3L) + masks
        # the border of the three planes written, the edge samples read (reference pictures only: the deepest layer is not padded)
        "pad": (pics_step // 2) * int((geo.pw * geo.ph - L) + 2 * (geo.cpw * geo.cph - L // 4) + 2 * (Hd + Wd)),
    }
    # the rocprofv3 name of the ME kernel the launches really were (svt_hip_me_last_instance: me_spec.h index, + 100 for csrc/me_fast.h's driver, + 200 for the compact-layout pair of launches)
    me_inst = int(lib.svt_hip_me_last_instance(me_ctxs[0])) if ctxs else -1
    me_kernel_name = (f"svt_me_sb_kernel<{me_inst - 200}, true>" if me_inst >= 200 else   # (the compact-layout launch; its <.., false> companion runs the flagged SBs)
                      f"svt_me_fast_kernel<{me_inst - 100}>" if me_inst >= 100 else f"svt_me_sb_kernel<{max(me_inst, 0)}, false>")
    kernel_of = {"pa": "svt_pa_plane_kernel", "me": me_kernel_name, "mc": "svt_mc_kernel", "lists": "svt_tq_count / svt_scan / svt_tq_emit kernels",
                 "tq": "svt_tq_kernel<4|8|16|32>", "skip": "svt_tq_skip / svt_skip_update kernels", "lf": "svt_lf_mask + svt_lf_desc + svt_lf_kernel",
                 "pad": "svt_refpad_kernel"}
    me_ms = max(stage_ms["me"], 1e-9)
    # roofline of the dominant kernel: algorithmic bytes of the step's ME launches / the time they take.  `frac` uses the clean
    # figure -- motion estimation alone on one stream (me_alone) when it was measured, else the busiest ME stream's span inside
    # the step; the per-launch durations inside the step are stretched by the launches running beside them and are given separately
    me_clean_ms = me_alone["launch_ms_sum_per_step"] if me_alone else me_ms
    achieved = stage_bytes["me"] / (me_clean_ms * 1e-3) / 1e9
    traffic, valu, traffic_source = None, None, None
    # the committed counter passes of this configuration (tools/profile_round.sh TAG PRESET): profiles/traffic.json is c3's
    tj_name = "traffic.json" if args.preset == "c3" else f"traffic_{args.preset}.json"
    tj = os.path.join(ROOT, "profiles", tj_name)
    if os.path.exists(tj) and (Wd, Hd) == PRESETS[args.preset][:2]:
        tjd = json.load(open(tj))
        rec = tjd.get("svt_me_kernel") or tjd.get("svt_me_sb_kernel", {})   # (the family's key until round 5)
        if rec.get("bytes_per_step"):   # the profiling run's step is one mini-GOP of one GOP
            traffic = int(rec["bytes_per_step"] * G / n_launch_step)      # per launch, like `achieved`
            traffic_source = f"profiles/{tj_name} (rocprofv3 --pmc passes of tools/profile_round.sh, committed; not measured in this run)"
        vi = rec.get("valu_wave_insts_per_step")
        if vi and "me" in stages:
            # the kernel's real roof: 64-lane VALU instructions issued (rocprofv3 SQ_INSTS_VALU, profiles/) per second against
            # 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz (MI355X_MICROARCH.md)
            ach = vi * G * 64 / (me_clean_ms * 1e-3) / 1e12
            valu = {"achieved": round(ach, 2), "peak": 39.3, "unit": "T lane-ops/s", "frac": round(ach / 39.3, 4), "wave_insts_per_minigop": vi,
                    "source": f"profiles/{tj_name}"}
    fps = GS.aggregate_rate(pics_step + n_keys_main / args.steps, args.steps, world, dt)   # every rank codes the same number of key frames
    stages_run = [s for s in STAGES if s in stages or (s in ED_STAGE_NAMES and run_encdec)]
    out = {
        "metric": "hot-path frames/sec (+ Mpixels/sec): block-level DSP path (picture analysis + ME + inter prediction from reconstructed references + "
                  f"DCT/quant/recon + deblock + reference padding), {Wd}x{Hd}p60 yuv420p enc-mode {enc_mode}; no mode decision, no entropy coding -- not encoded frames",
        "value": round(fps, 2),
        "unit": "frames/s",
        "mpixels_per_s": round(fps * Wd * Hd / 1e6, 1),
        "value_inter_only": round(inter_only["frames_per_s"], 2) if inter_only else None,
        "key_frames": None if not P_main["keys"] else {
            "per_step": round(n_keys_main / args.steps, 3), "intra_period": INTRA_PERIOD,
            "what": f"one key frame per closed GOP of {INTRA_PERIOD} inter pictures: its encode pass (svt_hip_encdec_intra_device: prediction + transform wavefront, "
                    "deblocking, border; stand-in decision 16x16 DC) runs inside the timed step on the GOP group's key stream, one step ahead of the mini-GOP "
                    "that predicts from it; `value` counts it as one frame, `value_inter_only` is the same step without key frames",
            "workgroups": int(os.environ.get("SVT_BENCH_KEY_WGS", "128")),   # svt_hip_ctx_set_intra_workgroups on the key streams (default of the library: one per CU)
            "alone_ms": round(key_alone_ms, 3) if key_alone_ms else None,
            "ms_per_step_with": round(dt / args.steps * 1e3, 3), "ms_per_step_without": round(inter_only["ms_per_step"], 3) if inter_only else None,
            "hidden_fraction": (round(1.0 - max(0.0, dt / args.steps * 1e3 - inter_only["ms_per_step"]) / max(1e-9, key_alone_ms * n_keys_main / args.steps), 3)
                                if (inter_only and key_alone_ms and n_keys_main) else None)},
        "value_reference_flags": round(ref_flags["frames_per_s"], 2) if ref_flags else None,
        "single_stream_value": round(single["diagonal"]["frames_per_s"], 2) if single else None,
        "single_gop_value": round(single["waves"]["frames_per_s"], 2) if single else None,
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3),
        "ms_per_minigop": round(dt / args.steps / G * 1e3, 3),
        "host_enqueue_ms_per_step": round(enq_s * 1e3, 3),
        "host_prepare_ms_per_step": 0.0,
        "host_prepare_note": "nothing is prepared on the host per picture: transform-block lists, skip flags and LOOP_FILTER_MASKs are built on the device "
                             "inside the timed region (stages `lists`, `skip`, `lf`); the mode-info grids (mode decision's output) are synthesised once in set-up",
        "setup_s": round(setup_s, 1),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u8",
        "data": "synthetic",
        "config": {"workload": f"{args.preset}: {Wd}x{Hd} 8-bit yuv420p, -enc-mode {enc_mode} -tune {tune} -q 40, 1xMI355X per rank; step = one {MINIGOP}-picture mini-GOP "
                               f"({n_layers} temporal layers, B pictures, 2 reference lists) of each of {G} closed GOPs in flight; ME side on source pictures one "
                               "mini-GOP ahead; EncDec side in dependency order (" +
                               ("per step temporal layer l of the mini-GOP l steps back: every picture is predicted from reconstructions finished in "
                                "earlier steps" if args.schedule == "diagonal" else "dependent temporal-layer waves per mini-GOP") +
                               "), per batch one svt_hip_encdec_batch_device call: inter prediction from the deblocked + padded reconstruction of the references "
                               "-> block lists from the mode-info grid (device) -> transform / quant / recon -> skip flags -> masks (device) -> deblocking -> "
                               "reference padding, in place in the reference buffers",
                   "preset": args.preset, "gops_in_flight": G, "gop_groups": n_groups, "schedule": args.schedule,
                   "stages": ["picture_analysis", "motion_estimation", "inter_prediction", "block_lists", "transform_quant_recon", "skip_flags",
                              "masks_and_deblocking", "reference_padding"],
                   "stages_run": stages_run,
                   "pictures_per_step": pics_step, "transform_blocks_per_step": n_blocks_step, "q_index": Q_INDEX,
                   "deblocked_pictures": "value: all (the reference's behaviour with reconstructed output enabled); value_reference_flags: the reference's flags "
                                         "without reconstructed output (svt_hip_encdec_flags_derive)",
                   "not_in_the_step": "coefficient rate (coeff_rate_estimate): mode decision calls it, the encode pass does not (Codec/EbEncDecProcess.c:700-745 vs "
                                      ":3627-4241); svt_hip_tq_rd_batch_device serves a host decision loop and is covered by the GPU tests, not by this step",
                   "workload_stats": workload_stats,
                   "parallelism": f"gop-shard x{world}" + (" + split-GOP reference hand-off (RCCL send/recv of the padded base-layer reconstruction)" if handoff else "")},
        "reference_flags": None if not ref_flags else {"frames_per_s": round(ref_flags["frames_per_s"], 2), "ms_per_minigop": round(ref_flags["ms_per_minigop"], 3),
                                                       "steps": ref_flags["steps"], "flags_by_layer": ref_flags["flags_by_layer"]},
        "single_stream": None if not single else {
            "frames_per_s": round(single["diagonal"]["frames_per_s"], 2), "ms_per_minigop": round(single["diagonal"]["ms_per_minigop"], 3), "gops_in_flight": 1,
            "schedule": "diagonal", "steps": single["diagonal"]["steps"],
            "stage_ms_per_minigop": {s: round(single["diagonal"]["stage_ms"][s], 3) for s in stages_run},
            "note": "ONE stream of mini-GOPs: per step temporal layer l of the mini-GOP l steps back (mutually independent pictures of consecutive "
                    "mini-GOPs, each predicted from reconstructions finished in earlier steps) -- the picture-level pipelining the reference's EncDec "
                    "processes do; needs as many mini-GOPs of look-ahead as there are temporal layers"},
        "single_gop": None if not single else {
            "frames_per_s": round(single["waves"]["frames_per_s"], 2), "ms_per_minigop": round(single["waves"]["ms_per_minigop"], 3), "gops_in_flight": 1,
            "schedule": "waves", "steps": single["waves"]["steps"],
            "stage_ms_per_minigop": {s: round(single["waves"]["stage_ms"][s], 3) for s in stages_run},
            "note": "one mini-GOP at a time: its temporal-layer waves run one after the other (only ME / picture analysis of the next "
                    "mini-GOP overlap them) -- the lowest-latency schedule, bounded by the per-picture latency of the deblocking wavefront"},
        "roofline": {"bound": "hbm", "kernel": me_kernel_name, "achieved": round(achieved, 2), "peak": 8000.0,
                     "unit": "GB/s", "frac": round(achieved / 8000.0, 5), "traffic": traffic, "traffic_source": traffic_source,
                     "launches_per_step": n_launch_step, "avg_launch_ms": round(me_clean_ms / n_launch_step, 4),
                     "algorithmic_bytes_per_launch": int(stage_bytes["me"] / n_launch_step),
                     "denominator": ("motion estimation alone on one stream: sum of the step's launch durations (HIP events on the launch stream)" if me_alone else
                                     "busiest ME stream's span inside the step (HIP events on the launch streams)"),
                     "in_step": {"busiest_stream_ms_per_step": round(me_ms, 3), "GB_per_s": round(stage_bytes["me"] / (me_ms * 1e-3) / 1e9, 2),
                                 "frac": round(stage_bytes["me"] / (me_ms * 1e-3) / 8e12, 5),
                                 "stretched_launch_ms_sum_per_step": round(sum(me_launch_list) / args.steps, 3),
                                 "note": f"{n_me_streams} ME streams run launches concurrently beside the EncDec streams: a launch's own duration is stretched, "
                                         "their sum can exceed the step; the busiest stream's span cannot"},
                     "me_alone": None if not me_alone else {k_: round(v_, 3) for k_, v_ in me_alone.items()},
                     "valu": valu},
        "kernels": {kernel_of[s]: {"ms_per_step": round(stage_ms[s], 3), "ms_per_step_all_streams": round(stage_ms_sum[s], 3), "algorithmic_bytes_per_step": stage_bytes[s],
                                   "GB_per_s": round(stage_bytes[s] / (max(stage_ms_sum[s], 1e-9) * 1e-3) / 1e9, 2),
                                   "frac_of_8TBps": round(stage_bytes[s] / (max(stage_ms_sum[s], 1e-9) * 1e-3) / 8e12, 5)}
                    for s in stages_run},
        "kernels_note": "ms_per_step = the busiest stream's total for that stage (<= ms_per_step); ms_per_step_all_streams = summed over the concurrent streams "
                        "(the GOP groups run side by side, so the sums can exceed the wall time); GB_per_s uses the sum",
        "pcie_note": f"inputs are resident in HBM when the clock starts; a {Wd}x{Hd} 4:2:0 picture is {pic_bytes / 1e6:.1f} MB, so {round(fps)} frames/s "
                     f"would need {fps * pic_bytes / 1e9:.0f} GB/s of host-to-device traffic if every picture crossed PCIe (gen5 x16 sustains ~50): the "
                     "PCIe-inclusive rate of the public-API path is `api_path` (app/svt_enc_api_bench.c, DESIGN.md section 7)",
    }
    cpu_inputs = None
    if not args.no_cpu_baseline and world == 1:   # the CPU leg runs at N = 1 only (rank 0's host cores are not shared with other ranks)
        lfm0 = {i: np.frombuffer(d_lfm[0, i - 1].cpu().numpy(), dtype=B.LF_MASK_DTYPE).reshape(sb_rows, sb_cols).copy() for i in range(1, MINIGOP + 1)}
        lf0 = {i: d_lf[0][i].cpu().numpy().view(B.LF_MODE_INFO_DTYPE).reshape(mi_rows, mi_cols).copy() for i in range(1, MINIGOP + 1)}   # with the skip flags
        ref_inputs = {"pred": d_pred[0, MINIGOP // 4 - 1].cpu().numpy(), "kcell": kcell[0][MINIGOP // 4], "level": level}
        cpu_inputs = (lf0, lfm0, ref_inputs)   # (the leg itself runs last: its processes would share the host's cores with the public-API run below)
    for grp in P_main["groups"] + (P_single["groups"] if P_single else []) + (P_ref["groups"] if P_ref else []) + (P_inter["groups"] if P_inter else []):
        lib.svt_hip_encdec_work_destroy(grp["ctx"], grp["work"])
        if grp["kres"]:
            lib.svt_hip_encdec_work_destroy(grp["ictx"], grp["kres"]["work"])
    for c_ in ctxs:
        lib.svt_hip_ctx_destroy(c_)
    ctxs.clear()
    torch.cuda.synchronize()
    del keep[:]
    torch.cuda.empty_cache()
    out.update(early_legs)
    if cpu_inputs is not None:
        lf0, lfm0, ref_inputs = cpu_inputs
        out["cpu_baseline"] = cpu_baseline(T, B, frames0, src0, mc_host0, lf0, lfm0, Wd, Hd, enc_mode, tune, l1_on, ref_inputs)
    print(json.dumps(out))


# -----------------------------------------------------------------------------------------------------------------------
# CPU baseline: the oracle (kind "port"), built -O3 -march=native on this host, one whole picture per PROCESS
# -----------------------------------------------------------------------------------------------------------------------
def _cpu_worker(job):
    """one whole picture through all the stages of the path with the oracle, single-threaded; returns the seconds per stage"""
    lib_path, data_path, i, Wd, Hd, enc_mode, tune, l1_on, levels = job
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.modules.setdefault("torch", None)     # the host-side helpers of the product library need no torch: keep 256 processes light
    import svt_testlib as T
    import encdec_model as M
    B = T.B
    set_structure(levels)
    orc = C.CDLL(lib_path)
    d = np.load(data_path, mmap_mode="r")
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    t = {}
    a, b = refs_of(i)
    lum = {j: np.ascontiguousarray(d["src"][j, :Wd * Hd].reshape(Hd, Wd)) for j in {i, a, b}}

    class HostPa:
        def __init__(self, luma):
            self.arr = [np.zeros(((Hd >> sh) + 2 * pd, (Wd >> sh) + 2 * pd), np.uint8) for sh, pd in ((0, 68), (1, 32), (2, 16))]
            self.d = B.PaPicture()
            self.d.full, self.d.quarter, self.d.sixteenth = (B.plane_desc(x, pd, pd) for x, pd in zip(self.arr, (68, 32, 16)))
            self.luma = luma
    pics = {j: HostPa(lum[j]) for j in lum}
    t0 = time.perf_counter()
    for j in pics:
        assert orc.svt_oracle_pa_prepare(vp(pics[j].luma), Wd, C.byref(pics[j].d), 1 if l1_on else 0) == 0
    t["pa"] = (time.perf_counter() - t0) / len(pics)      # one picture's share: every picture is analysed once
    p = B.me_params_preset(Wd, Hd, enc_mode, tune, 2, LAYER[i - 1], levels)
    res = np.zeros((T.n_sb(Wd, Hd), 85), dtype=B.ME_RESULT_DTYPE)
    t0 = time.perf_counter()
    assert orc.svt_oracle_me_picture(C.byref(pics[i].d), C.byref(pics[a].d), C.byref(pics[b].d), C.byref(p), vp(res), None, 0, -1) == 0
    t["me"] = time.perf_counter() - t0
    # the stages behind mode decision: the oracle chain (tests/encdec_model.py), references = the padded source pictures
    def planes(j):
        s = d["src"][j]
        return (np.ascontiguousarray(s[:Wd * Hd].reshape(Hd, Wd)), np.ascontiguousarray(s[Wd * Hd:Wd * Hd * 5 // 4].reshape(Hd // 2, Wd // 2)),
                np.ascontiguousarray(s[Wd * Hd * 5 // 4:].reshape(Hd // 2, Wd // 2)))
    refs = [M.RefPic(Wd, Hd).set_padded(*planes(j)) for j in (a, b)]
    thr = B.LfThresh()
    orc.svt_oracle_lf_thresh_init(C.byref(thr), 0)
    fl = B.EncdecFlags(limit_intra=0, allow_enc_dec_mismatch=0, do_recon=1, apply_loop_filter=1, pad_reference=1)
    M.oracle = lambda: orc           # the natively compiled oracle does the work
    T.oracle = lambda: orc
    times = {}
    M.oracle_encdec_picture(planes(i), refs, np.ascontiguousarray(d[f"mc{i}"]), np.ascontiguousarray(d[f"lf{i}"]), Q_INDEX, fl, thr, timings=times)
    t.update(times)
    return t


def cpu_baseline(T, B, frames, src_all, mc_mi, lf_mi, lfm, Wd, Hd, enc_mode, tune, l1_on, ref_inputs=None):
    """The oracle (C restatement of the reference's C path; kind "port"), compiled -O3 -march=native on this host, on the SAME
    workload: whole pictures of GOP 0's mini-GOP through all the stages of the step, ONE PICTURE PER PROCESS, as many processes as
    the sample has pictures -- the reference's own parallelism is pictures in flight x segments, and independent pictures are what
    lets every core work.  Wall-clock of the slowest process.  Timed with min(host cores, memory-bound cap) processes and with 8
    (SURVEY 8(d)).  Reported, never the target."""
    import multiprocessing as mp
    ncpu, hw_threads, quota = effective_cpus()
    lib_path = build_native_oracle()
    orc = C.CDLL(lib_path)
    try:
        import psutil
        mem_cap = max(1, int(psutil.virtual_memory().available // (Wd * Hd * 40 + (256 << 20))))
    except Exception:
        mem_cap = 64
    with tempfile.TemporaryDirectory() as td:
        data_path = os.path.join(td, "cpu_leg.npz")
        arrs = {"src": src_all}
        for i in mc_mi:
            arrs[f"mc{i}"], arrs[f"lf{i}"] = mc_mi[i], lf_mi[i]
        np.savez(data_path, **arrs)

        def run(n_proc):
            jobs = [(lib_path, data_path, 1 + (k % MINIGOP), Wd, Hd, enc_mode, tune, l1_on, LEVELS) for k in range(n_proc)]
            with mp.get_context("spawn").Pool(n_proc) as pool:
                pool.map(_cpu_warm, range(n_proc))            # processes up, numpy + the oracle loaded
                t0 = time.perf_counter()
                ts = pool.map(_cpu_worker, jobs, chunksize=1)
                wall = time.perf_counter() - t0
            stage = {s: float(np.mean([t_[s] for t_ in ts])) for s in ts[0]}
            return n_proc / wall, wall, stage
        n_all = max(1, min(ncpu, mem_cap, 256))
        fps_all, wall_all, stage_all = run(n_all)
        fps_8, wall_8, stage_8 = run(min(8, n_all)) if n_all > 8 else (fps_all, wall_all, stage_all)
    ref_stages = reference_stage_rates(T, B, src_all, ref_inputs["pred"], ref_inputs["kcell"], mc_mi[MINIGOP // 4], Wd, Hd, ref_inputs["level"], ncpu) if ref_inputs else None
    return {"value": round(fps_all, 3), "unit": "frames/s", "cores": n_all, "hardware_threads": hw_threads, "cpu_quota_cores": quota, "kind": "port", "cpu_model": cpu_model(),
            **(ref_stages or {}),
            "value_8_cores": round(fps_8, 3), "scaling_vs_8_cores": round(fps_all / max(fps_8, 1e-9), 2),
            "seconds_all_cores": round(wall_all, 2), "seconds_8_cores": round(wall_8, 2),
            "reference_me": reference_me_rate(T, B, orc, frames, Wd, Hd, enc_mode, tune, l1_on, ncpu),
            "stage_seconds_per_picture_1_core": {k: round(v, 3) for k, v in stage_8.items()},
            "stage_seconds_per_picture_all_cores_busy": {k: round(v, 3) for k, v in stage_all.items()},
            "sample": f"C-path proxy: oracle (C restatement of the reference's C path, gcc -O3 -march=native on this host = auto-vectorised \"AVX2-class\" "
                      f"proxy; the reference's AVX2 / yasm build cannot be made here) on {n_all} whole {Wd}x{Hd} pictures of the same mini-GOP through the "
                      f"stages of the step (picture analysis, ME, inter prediction, transform / quant / recon, skip flags, masks + deblocking, padding), one "
                      f"single-threaded process per picture, {n_all} processes at once (value) and 8 (value_8_cores); wall-clock of the slowest process.  "
                      f"cores = what this container may use: {hw_threads} hardware threads are visible, the cgroup CPU quota is "
                      f"{quota if quota else 'unlimited'} CPUs' worth of time (more busy processes than that only take turns)"}


def _cpu_warm(_k):
    sys.modules.setdefault("torch", None)
    import svt_testlib  # noqa: F401
    svt_testlib.B.load()
    return 0


if __name__ == "__main__":
    main()
