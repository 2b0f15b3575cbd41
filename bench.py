#!/usr/bin/env python3
"""bench.py -- hot-path throughput of the MI355X implementation of SVT-VP9's block-level DSP path, dependency-true.

One "step" = one 16-picture mini-GOP (5 temporal layers, hierarchical_levels = 4, B pictures with 2 reference lists) of EACH of
G concurrent closed GOPs (closed GOPs are independent streams of work) of synthetic 3840x2160 8-bit 4:2:0 input at the
enc-mode 8 / tune 1 (OQ) settings, q index 160 (-q 40), through the whole data path in the order the data dependencies of the
encoder impose:

    ME side (source pictures only, one mini-GOP ahead, own streams):
        picture analysis (padded + 1/16 planes from the source luma)  ->  motion estimation (16 B pictures per GOP)
    EncDec side, per GOP FIVE DEPENDENT WAVES -- temporal layer 0 (picture 16), 1 (8), 2 (4, 12), 3 (2, 6, 10, 14), 4 (odd) --
    each wave:
        inter prediction FROM THE DEBLOCKED, PADDED RECONSTRUCTION of its lower-layer reference pictures (svt_mc_kernel)
        ->  residual / transform / quantisation / reconstruction + distortion + coefficient rate (one fused pass) into the
            picture's reference buffer  ->  in-loop deblocking in place  ->  border padding in place (svt_refpad_kernel):
            the picture is now the reference of the next wave (and picture 16 the base of the next mini-GOP)

A wave of layer l of a mini-GOP cannot start before the wave of layer l - 1 has been padded: the reconstruction buffers ARE the
reference pictures (as in the reference, Codec/EbEncDecProcess.c:4822-4851, 5676-5696) and every stage runs on the stream of
its GOP group in program order.  GOPs are split into groups that run on separate streams, so the deblocking tail of one group
overlaps the transform work of another.  `value` uses G GOPs in flight (config.gops_in_flight; enough to fill the GPU);
`single_gop_value` is the same path with G = 1 (one stream of mini-GOPs, nothing to overlap the dependent waves with).

The mode-info grids (partition, prediction direction and motion vectors of every block) are mode decision's output -- host
logic outside this path -- and are built ONCE before the timed loop from a first ME pass over the same pictures, as are the
loop-filter masks (svt_hip_lf_build_masks on the same grids, skip flags from a first pass of the dependent chain).  Deblocking
runs on every picture (the reference does so when it writes reconstructed output, `-o`; without it, it skips the pictures that are
not used as references).  Inputs are resident in HBM before the timed region.  value = pictures / second (whole job, all ranks).

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
PAD = 80        # border of a reference picture: 64 + 16 (Codec/EbEncHandle.c:968-971); chroma half of it
# temporal layer of picture i (1..16) inside a 16-picture mini-GOP (5 layers, hierarchical_levels = 4)
LAYER = [4, 3, 4, 2, 4, 3, 4, 1, 4, 3, 4, 2, 4, 3, 4, 0]
STAGES = ("pa", "me", "mc", "tq", "rate", "lf", "pad")   # "rate" is a launch of its own only with SVT_BENCH_SEPARATE_RATE=1 (A/B aid)


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




def api_path_rate(frames, Wd, Hd, n_send=130):
    """SURVEY 8(d)'s metric through the PUBLIC API (libSvtVp9Enc.so, the reference's eb_vp9_svt_* entry points): wall-clock from the
    first send_picture to the EOS packet, host buffers handed over (PCIe inside the clock), measured by the plain-C caller
    app/svt_enc_api_bench.c in its own process.  Behind the API the library runs picture analysis + one batched ME launch per
    mini-GOP + the per-SB ME statistics; no other stage is reachable from it, and only the luma plane
    crosses PCIe.  n_send = 130: two closed GOPs of 65 pictures at 60 frames/s."""
    exe = os.path.join(ROOT, "app", "svt_enc_api_bench")
    if not os.path.exists(exe):
        return None
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "luma.bin")
        with open(path, "wb") as f:
            for y in frames:
                f.write(np.ascontiguousarray(y).tobytes())
        r = subprocess.run([exe, path, str(Wd), str(Hd), str(len(frames)), str(n_send), "8", "1"], capture_output=True, text=True)
    if r.returncode != 0:
        return {"error": f"svt_enc_api_bench rc={r.returncode}: {(r.stdout + r.stderr).strip()[-200:]}"}
    d = json.loads(r.stdout.strip().splitlines()[-1])
    return {"value": d["frames_per_s"], "unit": "frames/s", "frames": d["frames"], "seconds": d["seconds"], "me_launches": d["me_launches"],
            "mpixels_per_s": round(d["frames_per_s"] * Wd * Hd / 1e6, 1),
            "what": f"{Wd}x{Hd} -enc-mode 8 -tune 1, {d['frames']} pictures through eb_vp9_svt_enc_send_picture / eb_vp9_svt_get_packet (app/svt_enc_api_bench.c): "
                    "first send_picture -> EOS packet, host buffers in, PCIe included; stages behind the API: picture analysis + motion estimation (one "
                    "batched launch per mini-GOP) + per-SB ME statistics; zero-byte packets (no entropy coding)"}


def reference_me_rate(T, B, orc, frames, Wd, Hd, l1_on, ncpu):
    """The REFERENCE's own motion_estimate_sb (oracle/_ref/ref_me_sb = Codec/EbMotionEstimation.c compiled from /root/reference in
    the build container, C path, gcc -O2; it travels to the GPU box as a prebuilt file) timed beside the port on the same
    picture: one 4K B picture of temporal layer 2 of the mini-GOP, SB ranges over up to 64 processes / threads; the seconds
    are the SB loops' own (clock_gettime inside the harness: no request I/O).  Returns None when the prebuilt reference is
    absent."""
    from concurrent.futures import ThreadPoolExecutor
    import struct
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_me_sb")
    if not os.path.exists(exe):
        return None
    i = 4
    a, b = refs_of(i)
    pics = [T.PaPic(frames[j]) for j in (i, a, b)]
    p = B.me_params_preset(Wd, Hd, 8, 1, 2, LAYER[i - 1], 4)
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
            "sample": f"{n_done} of the {nsb} superblocks of one {Wd}x{Hd} B picture (temporal layer 2), {len(ranges)} processes x {per} SBs",
            "workers": len(ranges),
            "sb_per_s": round(n_done / max(t_ for t_, _ in ref_out), 1), "frames_per_s_me_only": round(n_done / max(t_ for t_, _ in ref_out) / nsb, 3),
            "sb_per_s_1_thread": round(n0 / one[0], 1),
            "port_sb_per_s": round(n_done / max(port_t), 1), "port_sb_per_s_1_thread": round(n0 / port_one, 1),
            "port_over_reference_1_thread": round((n0 / port_one) / (n0 / one[0]), 3),
            "results_identical": bool(same)}


class Geometry:
    """Layouts in HBM.  Source / prediction pictures: Y rows followed by U | V rows (half width each) at the luma stride.
    Reconstruction = reference pictures: three padded planes one after the other (Y with PAD samples of border, U and V with
    PAD / 2), as the reference allocates its reference pictures (Codec/EbEncHandle.c:968-971)."""

    def __init__(self, w, h):
        self.w, self.h = w, h
        self.plane_w, self.yuv_rows = w, h + h // 2
        self.pic_bytes = self.yuv_rows * self.plane_w
        self.pw, self.ph = w + 2 * PAD, h + 2 * PAD
        self.cpw, self.cph = w // 2 + PAD, h // 2 + PAD
        self.u_base = self.pw * self.ph
        self.v_base = self.u_base + self.cpw * self.cph
        self.rec_bytes = (self.v_base + self.cpw * self.cph + 63) // 64 * 64
        self.y0 = PAD * self.pw + PAD                               # offsets of sample (0,0) of each plane inside a reference picture
        self.u0 = self.u_base + (PAD // 2) * self.cpw + PAD // 2
        self.v0 = self.v_base + (PAD // 2) * self.cpw + PAD // 2
        self.coeffs = w * h * 3 // 2                                # transform coefficients of a picture

    def recon_offsets(self, tight_off):
        """offset of a transform block inside a padded reference picture (and its row stride) from its offset in the tight layout"""
        r, c = tight_off // self.plane_w, tight_off % self.plane_w
        luma = r < self.h
        isv = c >= self.w // 2
        cr, cc = r - self.h, np.where(isv, c - self.w // 2, c)
        off = np.where(luma, self.y0 + r * self.pw + c, np.where(isv, self.v0, self.u0) + cr * self.cpw + cc)
        return off.astype(np.int64), np.where(luma, self.pw, self.cpw).astype(np.uint16)


def pics_of_layer(layer):
    return [i for i in range(1, MINIGOP + 1) if LAYER[i - 1] == layer]


RING = 6   # mini-GOPs whose reference pictures are alive at a time (the diagonal schedule reaches back five mini-GOPs)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--width", type=int, default=W4K)
    ap.add_argument("--height", type=int, default=H4K)
    ap.add_argument("--gops", type=int, default=int(os.environ.get("SVT_BENCH_GOPS", "4")), help="closed GOPs in flight per GPU (value); single_gop_value always uses 1")
    ap.add_argument("--groups", type=int, default=int(os.environ.get("SVT_BENCH_GROUPS", "2")), help="GOP groups = EncDec streams")
    ap.add_argument("--schedule", choices=("diagonal", "waves"), default=os.environ.get("SVT_BENCH_SCHEDULE", "diagonal"),
                    help="EncDec-side schedule of `value`: diagonal = per step, temporal layer l of the mini-GOP l steps back (one batch of mutually "
                         "independent pictures per GOP group); waves = the five layers of the newest mini-GOP one after the other")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-single", action="store_true", help="skip the G = 1 run (single_gop_value)")
    ap.add_argument("--handoff", action="store_true", help="N > 1 only: split-GOP mode -- every step each rank also hands the padded base-layer reconstruction of its "
                    "first GOP to the next rank (RCCL send / recv over xGMI), the one exchange step of the path; closed GOPs (the default) need none")
    ap.add_argument("--stages", default=",".join(STAGES), help="profiling aid: run only these stages (the contract run uses all)")
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
    import me_configs as MC  # noqa: F401
    import svt_testlib as T
    _sp = importlib.util.spec_from_file_location("gop_shard", os.path.join(ROOT, "svt-vp9_amd", "gop_shard.py"))
    GS = importlib.util.module_from_spec(_sp)
    _sp.loader.exec_module(GS)
    B = T.B
    lib = B.load()
    dev = torch.device("cuda", local_rank)
    G = max(1, args.gops)
    n_groups = max(1, min(args.groups, G))
    t_setup0 = time.perf_counter()

    # streams: 0 = ME (+ a second ME stream: ME of different pictures is independent, the tail of one launch is filled by the
    # other), one per GOP group for the EncDec side (prediction -> transform -> deblocking -> padding, in program order = in
    # dependency order), one for picture analysis.  ME fills every CU by itself (5 workgroups use all of a CU's LDS and 480 of
    # the 512 registers per SIMD): the EncDec streams get the higher priority, so that when an ME workgroup retires a waiting
    # transform workgroup moves in first.
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
    single_pairs = [new_ctx(prio[1]) for _ in range(max(1, int(os.environ.get("SVT_BENCH_SINGLE_STREAMS", "2"))))]

    Wd, Hd = args.width, args.height
    geo = Geometry(Wd, Hd)
    nsbx = (Wd + 63) // 64
    nsb = T.n_sb(Wd, Hd)
    mi_rows, mi_cols = Hd // 8, Wd // 8
    sb_rows, sb_cols = (mi_rows + 7) // 8, (mi_cols + 7) // 8
    plane_w, yuv_rows, pic_bytes = geo.plane_w, geo.yuv_rows, geo.pic_bytes
    keep = []  # keeps device tensors alive

    def to_dev(a):
        t = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        keep.append(t)
        return t

    def dev_zeros(shape, dtype):
        t = torch.zeros(shape, dtype=dtype, device=dev)
        keep.append(t)
        return t

    # ---- synthetic input, resident in HBM: G GOP segments of 17 pictures (index 0 = the previous mini-GOP's base picture) ----
    d_src = dev_zeros((G, MINIGOP + 1, yuv_rows, plane_w), torch.uint8)
    frames_all, src0 = [], None
    for g in range(G):
        frames = T.gen_clip(Wd, Hd, MINIGOP + 1, seed=GS.gop_seed(11, rank * G + g))   # rank r encodes its own GOP segments
        src_all = np.zeros((MINIGOP + 1, yuv_rows, plane_w), np.uint8)
        for i, y in enumerate(frames):
            src_all[i, :Hd] = y
            src_all[i, Hd:, :Wd // 2] = y[::2, ::2] // 2 + 32
            src_all[i, Hd:, Wd // 2:] = 255 - y[::2, ::2] // 2 - y[1::2, 1::2] // 4
        d_src[g].copy_(torch.from_numpy(src_all))
        frames_all.append(frames if g == 0 else None)   # the CPU baseline works on GOP 0
        if g == 0:
            src0 = src_all
    src_ptr = lambda g, i: d_src.data_ptr() + (g * (MINIGOP + 1) + i) * pic_bytes

    # ---- stage "pa": the three ME planes of every picture from its luma (EbPaReferenceObject planes: padding 68 / 32 / 16) ----
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
    pa_sets = [[[pa_alloc() for _ in range(MINIGOP + 1)] for _ in range(G)] for _ in range(2)]

    def pa_call(ctx_, gops, idx, s=0):
        items = [(g, i) for g in gops for i in idx]
        n = len(items)
        lum = (C.c_void_p * n)(*[src_ptr(g, i) for g, i in items])
        strides = (C.c_int32 * n)(*[plane_w] * n)
        out = (B.PaPicture * n)(*[pa_sets[s][g][i] for g, i in items])
        B.check(lib.svt_hip_pa_prepare_batch_device(ctx_, n, lum, strides, out, 1 if l1_on else 0))

    all_gops = list(range(G))
    for s_ in range(2):
        pa_call(me_ctxs[0], all_gops, [0], s_)   # the previous mini-GOP's base picture: analysed when that mini-GOP was
    B.check(lib.svt_hip_ctx_synchronize(me_ctxs[0]))
    pa_idx = list(range(1, MINIGOP + 1))

    # ---- stage "me": one batched launch per temporal layer over the GOPs of the pipeline ----
    results = [[dev_zeros((nsb, 85 * 10), torch.int32) for _ in range(MINIGOP + 1)] for _ in range(G)]
    per_layer = not os.environ.get("SVT_BENCH_ME_ONE_LAUNCH")

    def build_me_launches(gops):
        sets = []
        for s in range(2):
            launches = []
            if per_layer:
                groups = [pics_of_layer(layer) for layer in range(5)]
            else:   # the whole step in one launch, biggest search areas (lowest layers) first
                groups = [sorted(range(1, MINIGOP + 1), key=lambda i: LAYER[i - 1])]
            for idx in groups:
                items = [(g, i) for g in gops for i in idx]
                n = len(items)
                p = (B.MeParams * n)()
                for k_, (g, i) in enumerate(items):
                    p[k_] = B.me_params_preset(Wd, Hd, 8, 1, 2, LAYER[i - 1], 4)
                    p[k_].same_ref_poc = 1 if LAYER[i - 1] == 0 else 0
                cur = (B.PaPicture * n)(*[pa_sets[s][g][i] for g, i in items])
                r0 = (B.PaPicture * n)(*[pa_sets[s][g][refs_of(i)[0]] for g, i in items])
                r1 = (B.PaPicture * n)(*[pa_sets[s][g][refs_of(i)[1]] for g, i in items])
                res = (C.c_void_p * n)(*[results[g][i].data_ptr() for g, i in items])
                launches.append((n, cur, r0, r1, p, res))
            sets.append(launches)
        # launches -> ME streams: largest first onto the least loaded stream
        slot, load = [0] * len(sets[0]), [0] * len(me_ctxs)
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

        def pair(self):
            return self.ev.pop(), self.ev.pop()

    def run_me(P, s, pool=None):
        for st_ in me_streams[1:]:
            st_.wait_stream(me_streams[0])
        for li, (n, cur, r0, r1, p, res) in enumerate(P["me_sets"][s]):
            k_ = P["me_slot"][li]
            if pool is not None:
                e0, e1 = pool.pair()
                e0.record(me_streams[k_])
            B.check(lib.svt_hip_me_batch_layers_device(me_ctxs[k_], n, cur, r0, r1, p, res, None))
            if pool is not None:
                e1.record(me_streams[k_])
                P["me_ev"].append((e0, e1))
        for st_ in me_streams[1:]:
            me_streams[0].wait_stream(st_)

    # ---- first pass (setup, untimed): PA + ME of every GOP, results to the host = the input of the synthesised mode decision ----
    P_all = {"gops": all_gops}
    P_all["me_sets"], P_all["me_slot"] = build_me_launches(all_gops)
    P_all["me_ev"] = []
    pa_call(me_ctxs[0], all_gops, pa_idx)
    with torch.cuda.stream(me_streams[0]):
        run_me(P_all, 0)
    for c_ in ctxs:
        B.check(lib.svt_hip_ctx_synchronize(c_))
    torch.cuda.synchronize()

    area_rows, area_cols = (Hd + 31) // 32, (Wd + 31) // 32
    iscan, ioffs = T.iscan_array()
    iscan_off_dct = [ioffs[(ts, 0)] for ts in range(4)]
    rtab, rscan = T.rate_tables()
    d_mi = [[None] * (MINIGOP + 1) for _ in range(G)]
    blk_tight = [[None] * (MINIGOP + 1) for _ in range(G)]     # per picture: the four per-size block arrays, offsets relative to the picture
    kcell = [[None] * (MINIGOP + 1) for _ in range(G)]
    mi_host0 = []
    inter_units = comp_units = mi_units = 0
    for g in range(G):
        rng = np.random.default_rng(5 + rank * G + g)
        for i in range(1, MINIGOP + 1):
            res = results[g][i].cpu().numpy().view(B.ME_RESULT_DTYPE).reshape(nsb, 85)
            kinds = rng.integers(0, 4, (area_rows, area_cols))
            if Hd % 32:   # blocks must not reach below the picture
                kinds[-1] = np.minimum(kinds[-1], 2 if Hd % 32 == 16 else 1)
            mi, k_cell = build_mode_info(B, res, kinds, mi_rows, mi_cols, nsbx)
            d_mi[g][i] = to_dev(mi.view(np.uint8))
            kcell[g][i] = k_cell
            blk_tight[g][i] = build_tq_blocks(B, kinds, Wd, Hd, plane_w, iscan_off_dct)
            inter, comp = mi["ref_list"][..., 0] >= 0, mi["ref_list"][..., 1] >= 0
            inter_units += int(inter.sum())
            comp_units += int((inter & comp).sum())
            mi_units += inter.size
            if g == 0:
                mi_host0.append(mi)

    # quantiser tables of q index 160, luma and chroma (no chroma deltas): the steps are the reference's eb_vp9_dc_quant /
    # eb_vp9_ac_quant values (committed fixture), the tables svt_hip_quant_tables_init's
    qrow = np.load(os.path.join(T.GOLDEN_DIR, "quant_reference.npz"))["0|0|0"][Q_INDEX]
    qtabs = np.zeros(2, dtype=B.QUANT_DTYPE)
    for j, base in enumerate((2, 14)):
        B.check(lib.svt_hip_quant_tables_init(Q_INDEX, int(qrow[1]), int(qrow[base]), int(qrow[base + 1]), qtabs[j:j + 1].ctypes.data_as(C.c_void_p)))
    ac_q = int(qrow[3])
    d_qt, d_iscan = to_dev(qtabs.view(np.uint8)), to_dev(iscan)
    d_rt, d_rs = to_dev(np.ascontiguousarray(rtab).reshape(1).view(np.uint8)), to_dev(rscan)

    # ---- EncDec-side arenas: prediction pictures, reference pictures (= reconstruction buffers), coefficients ----
    # Reference pictures live in a ring of RING mini-GOPs per GOP: picture i (1..16) of mini-GOP m is rec[m % RING][g][i - 1]; its
    # "picture 0" -- the base picture of the mini-GOP before -- is rec[(m - 1) % RING][g][15].  Nothing is ever copied: a picture is
    # reconstructed, deblocked and padded in the buffer later pictures predict from.
    d_pred = dev_zeros((G, MINIGOP, yuv_rows, plane_w), torch.uint8)
    d_rec = dev_zeros((RING, G, MINIGOP, geo.rec_bytes), torch.uint8)
    d_q, d_dq = dev_zeros(G * MINIGOP * geo.coeffs, torch.int16), dev_zeros(G * MINIGOP * geo.coeffs, torch.int16)
    slot_bytes = G * MINIGOP * geo.rec_bytes
    assert max(d_src.numel(), slot_bytes, d_q.numel()) < 2 ** 32, "svt_tq_block offsets are 32 bits: fewer GOPs in flight"
    rec_ptr = lambda slot, g, i: d_rec.data_ptr() + ((slot % RING) * G * MINIGOP + g * MINIGOP + i - 1) * geo.rec_bytes
    pred_ptr = lambda g, i: d_pred.data_ptr() + (g * MINIGOP + i - 1) * pic_bytes

    def yuv_desc(d, slot, g, i):
        base = rec_ptr(slot, g, i)
        d.y, d.u, d.v = base + geo.y0, base + geo.u0, base + geo.v0
        d.y_stride, d.uv_stride, d.width, d.height = geo.pw, geo.cpw, Wd, Hd

    def ref_desc(d, slot, g, j):
        """reference picture j (0..16) of the mini-GOP in ring slot `slot`"""
        if j == 0:
            yuv_desc(d, slot - 1, g, MINIGOP)
        else:
            yuv_desc(d, slot, g, j)

    # the base picture of the mini-GOP before the first one (ring slot -1): its source, padded -- from then on every reference is
    # a reconstruction
    for g in range(G):
        s0 = d_src[g, 0]
        y, u, v = s0[:Hd], s0[Hd:, :Wd // 2], s0[Hd:, Wd // 2:]
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

    level = lib.svt_hip_lf_level_from_q(ac_q, 0)
    thr = B.LfThresh()
    lib.svt_hip_lf_thresh_init(C.byref(thr), 0)
    d_lfm = [[dev_zeros(sb_rows * sb_cols * B.LF_MASK_DTYPE.itemsize, torch.uint8) if i else None for i in range(MINIGOP + 1)] for _ in range(G)]
    separate_rate = os.environ.get("SVT_BENCH_SEPARATE_RATE", "0") == "1"
    rate_ctx_rng = np.random.default_rng(8)

    def build_layer_blocks(items):
        """transform blocks of the pictures (g, i) of one temporal layer of a group, grouped by transform size; reconstruction offsets
        are relative to the ring slot the launch is given"""
        lb = {"items": items, "counts": [0, 0, 0, 0], "blocks_host": np.zeros(0, dtype=B.TQ_BLOCK_DTYPE), "pic_of_block": np.zeros(0, np.int32)}
        if not items:
            return lb
        per_ts, pic_of = [[] for _ in range(4)], [[] for _ in range(4)]
        for k, (g, i) in enumerate(items):
            arrs = blk_tight[g][i]
            nn_pic = np.concatenate([np.full(len(a), 16 << (2 * ts), np.int64) for ts, a in enumerate(arrs)])
            coff = np.concatenate([[0], np.cumsum(nn_pic)[:-1]]) + (g * MINIGOP + i - 1) * geo.coeffs
            pos = 0
            for ts, a in enumerate(arrs):
                b = a.copy()
                b["coeff_off"] = coff[pos:pos + len(a)].astype(np.uint32)
                pos += len(a)
                per_ts[ts].append(b)
                pic_of[ts].append(np.full(len(a), k, np.int32))
        blocks = np.concatenate([b for ts in range(4) for b in per_ts[ts]])
        lb["pic_of_block"] = np.concatenate([p_ for ts in range(4) for p_ in pic_of[ts]])
        lb["counts"] = [sum(len(b) for b in per_ts[ts]) for ts in range(4)]
        lb["cnt_c"] = (C.c_int32 * 4)(*lb["counts"])
        tight = blocks["src_off"].astype(np.int64)      # offsets relative to the picture, tight layout
        gi = np.array(items, np.int64)[lb["pic_of_block"]]
        roff, rstride = geo.recon_offsets(tight)
        blocks["src_off"] = (tight + (gi[:, 0] * (MINIGOP + 1) + gi[:, 1]) * pic_bytes).astype(np.uint32)
        blocks["pred_off"] = (tight + (gi[:, 0] * MINIGOP + gi[:, 1] - 1) * pic_bytes).astype(np.uint32)
        blocks["recon_off"] = (roff + (gi[:, 0] * MINIGOP + gi[:, 1] - 1) * geo.rec_bytes).astype(np.uint32)
        blocks["recon_stride"] = rstride
        rate_ctx = rate_ctx_rng.integers(0, 3, len(blocks)).astype(np.uint8)    # entropy context of every block (an input)
        blocks["pad"][:, 0] = rate_ctx | (blocks["qtab"] << 2) | (1 << 3)         # SVT_TQ_RATE_INFO(ctx, plane_type, is_inter = 1)
        lb["blocks_host"], lb["blocks"] = blocks, to_dev(blocks.view(np.uint8))
        nb = len(blocks)
        lb["eob"], lb["dist"], lb["bits"] = dev_zeros(nb, torch.int16), dev_zeros(2 * nb, torch.int64), dev_zeros(nb, torch.int32)
        if separate_rate:
            roffs, _ = T.rate_scan_offsets()
            rb = np.zeros(nb, dtype=B.RATE_BLOCK_DTYPE)
            rb["coeff_off"], rb["tx_size"], rb["plane_type"], rb["is_inter"], rb["ctx"] = blocks["coeff_off"], blocks["tx_size"], blocks["qtab"], 1, rate_ctx
            rb["scan_off"] = np.array([roffs[(ts, 0)] for ts in range(4)], np.uint32)[blocks["tx_size"]]
            lb["rb_host"] = rb
        return lb

    def build_diag_blocks(layers):
        """the transform blocks of the five temporal layers of a group in ONE batch, grouped by transform size across the layers; a
        block names its layer's reconstruction buffer (= ring slot) in pad_[0] bits 4-6 (svt_hip_tq_rd_batch_multi_device)"""
        per_ts = [[] for _ in range(4)]
        for layer, lb in enumerate(layers):
            b = lb["blocks_host"].copy()
            if len(b):
                b["pad"][:, 0] |= np.uint8(layer << 4)
            pos = 0
            for ts in range(4):
                per_ts[ts].append(b[pos:pos + lb["counts"][ts]])
                pos += lb["counts"][ts]
        blocks = np.concatenate([a for ts in range(4) for a in per_ts[ts]])
        counts = [sum(len(a) for a in per_ts[ts]) for ts in range(4)]
        nb = len(blocks)
        return {"blocks": to_dev(blocks.view(np.uint8)), "counts": counts, "cnt_c": (C.c_int32 * 4)(*counts), "n": nb,
                "eob": dev_zeros(nb, torch.int16), "dist": dev_zeros(2 * nb, torch.int64), "bits": dev_zeros(nb, torch.int32)}

    def build_batch(items):
        """descriptors of one batch of mutually independent pictures: items = (g, i, back) -- picture i of GOP g of the mini-GOP
        `back` mini-GOPs before the newest one of the step; one descriptor set per ring phase"""
        n = len(items)
        bt = {"n": n, "items": items, "mc": [], "yuv": []}
        if not n:
            return bt
        for ph in range(RING):
            mc = (B.McPicture * n)()
            yv = (B.YuvPlanes * n)()
            for k, (g, i, back) in enumerate(items):
                slot = ph - back
                mp = mc[k]
                mp.d_mi, mp.mi_stride, mp.mi_rows, mp.mi_cols, mp.use_subpel = d_mi[g][i].data_ptr(), mi_cols, mi_rows, mi_cols, 1
                for l in range(2):
                    ref_desc(mp.ref[l], slot, g, refs_of(i)[l])
                pb = pred_ptr(g, i)
                mp.pred.y, mp.pred.u, mp.pred.v = pb, pb + Hd * plane_w, pb + Hd * plane_w + Wd // 2
                mp.pred.y_stride, mp.pred.uv_stride, mp.pred.width, mp.pred.height = plane_w, plane_w, Wd, Hd
                yuv_desc(yv[k], slot, g, i)
            bt["mc"].append(mc)
            bt["yuv"].append(yv)
        bt["lfm"] = (C.c_void_p * n)(*[d_lfm[g][i].data_ptr() for g, i, _ in items])
        i32 = lambda v: (C.c_int32 * n)(*[v] * n)
        bt["lfs"], bt["mrs"], bt["mcs"] = i32(sb_cols), i32(mi_rows), i32(mi_cols)
        return bt

    vp = lambda t: C.c_void_p(t.data_ptr())
    slot_base = lambda slot: C.c_void_p(d_rec.data_ptr() + (slot % RING) * slot_bytes)

    def run_mc(ctx_, bt, ph):
        if bt["n"]:
            B.check(lib.svt_hip_inter_pred_batch_device(ctx_, bt["n"], bt["mc"][ph]))

    def run_tq(ctx_, lb, slot, plain=False):
        if not lb["items"]:
            return
        if plain or separate_rate:
            B.check(lib.svt_hip_tq_batch_dist_device(ctx_, vp(d_src), vp(d_pred), slot_base(slot), vp(lb["blocks"]), lb["cnt_c"], vp(d_qt), vp(d_iscan), vp(d_q),
                                                     vp(d_dq), vp(lb["eob"]), vp(lb["dist"])))
        else:   # distortion + rate behind the quantiser: perform_dist_rate_calc in one pass
            B.check(lib.svt_hip_tq_rd_batch_device(ctx_, vp(d_src), vp(d_pred), slot_base(slot), vp(lb["blocks"]), lb["cnt_c"], vp(d_qt), vp(d_iscan), vp(d_q),
                                                   vp(d_dq), vp(lb["eob"]), vp(lb["dist"]), vp(d_rt), vp(d_rs), vp(lb["bits"])))

    multi_tq = os.environ.get("SVT_BENCH_TQ_PER_LAYER", "0") != "1" and not separate_rate

    def run_tq_diag(ctx_, grp, ph):
        """the transform stage of a diagonal batch: one launch per transform size over all five layers, each layer reconstructing
        into the reference buffers of its own ring slot"""
        db = grp["diag_blocks"]
        rset = (C.c_void_p * 5)(*[d_rec.data_ptr() + ((ph - layer) % RING) * slot_bytes for layer in range(5)])
        B.check(lib.svt_hip_tq_rd_batch_multi_device(ctx_, vp(d_src), vp(d_pred), rset, 5, vp(db["blocks"]), db["cnt_c"], vp(d_qt), vp(d_iscan), vp(d_q), vp(d_dq),
                                                     vp(db["eob"]), vp(db["dist"]), vp(d_rt), vp(d_rs), vp(db["bits"])))

    def run_rate(ctx_, lb):
        B.check(lib.svt_hip_coeff_rate_batch_device(ctx_, vp(d_q), vp(lb["d_rb"]), len(lb["rb_host"]), vp(d_rt), vp(d_rs), vp(lb["bits"])))

    def run_lf(ctx_, bt, ph):
        if bt["n"]:
            B.check(lib.svt_hip_lf_batch_device(ctx_, bt["n"], bt["yuv"][ph], bt["lfm"], bt["lfs"], C.byref(thr), bt["mrs"], bt["mcs"], 0))

    def run_pad(ctx_, bt, ph):
        if bt["n"]:
            B.check(lib.svt_hip_ref_pad_batch_device(ctx_, bt["n"], bt["yuv"][ph], PAD, PAD))

    def build_pipeline(gops, pairs, split="gop"):
        """a pipeline = a set of GOPs in flight whose pictures are split into groups with one EncDec stream each: whole GOPs per
        group ("gop"), or -- any partition of a batch of mutually independent pictures is valid -- the pictures of every GOP dealt
        round-robin ("picture": what one stream of mini-GOPs uses to overlap the deblocking of one half of a batch with the transform
        stage of the other).  Per group: the transform block lists per temporal layer, the five wave batches (the group's pictures
        of one layer of the newest mini-GOP) and the diagonal batch (layer l of the mini-GOP l steps back)"""
        ng = len(pairs)
        if split == "gop":
            members = [[(g, i) for g in gops[k::ng] for i in range(1, MINIGOP + 1)] for k in range(ng)]
        else:
            members = [[(g, i) for g in gops for i in range(1, MINIGOP + 1) if i % ng == k] for k in range(ng)]
        P = {"gops": gops, "groups": [], "me_ev": []}
        # with the pictures of a GOP spread over several streams, a batch's references may have been finished on another stream:
        # every group then waits for the events all groups recorded behind their previous batch (events from a pool made here)
        P["cross_sync"] = split == "picture" and ng > 1
        P["sync_pool"] = [torch.cuda.Event() for _ in range(64 * ng)] if P["cross_sync"] else []
        for e_ in P["sync_pool"]:
            e_.record(me_streams[0])
        P["sync_pos"], P["sync_last"] = 0, []
        P["me_sets"], P["me_slot"] = build_me_launches(gops)
        for pics_, (st_, ctx_) in zip(members, pairs):
            by_layer = [[(g, i) for g, i in pics_ if LAYER[i - 1] == layer] for layer in range(5)]
            layers = [build_layer_blocks(by_layer[layer]) for layer in range(5)]
            P["groups"].append({"pics": pics_, "stream": st_, "ctx": ctx_, "layers": layers, "diag_blocks": build_diag_blocks(layers),
                                "waves": [build_batch([(g, i, 0) for g, i in by_layer[layer]]) for layer in range(5)],
                                "diag": build_batch([(g, i, LAYER[i - 1]) for g, i in pics_])})
        return P

    P_main = build_pipeline(all_gops, grp_pairs)

    # ---- setup pass of the dependent chain (untimed, mini-GOP 0 = ring phase 0, wave schedule): per wave prediction -> transform
    # -> [eobs -> loop-filter masks of the wave's pictures: skip flags are mode decision's output] -> deblocking -> padding ----
    eob_stats, resid_acc = [[] for _ in range(4)], []
    for layer in range(5):
        for grp in P_main["groups"]:
            bt, lb, ctx_ = grp["waves"][layer], grp["layers"][layer], grp["ctx"]
            if not bt["n"]:
                continue
            with torch.cuda.stream(grp["stream"]):
                run_mc(ctx_, bt, 0)
                run_tq(ctx_, lb, 0, plain=True)
            B.check(lib.svt_hip_ctx_synchronize(ctx_))
            eob_h = lb["eob"].cpu().numpy().view(np.uint16)
            blocks = lb["blocks_host"]
            if separate_rate:
                lb["rb_host"]["eob"] = eob_h
                lb["d_rb"] = to_dev(lb["rb_host"].view(np.uint8))
            for ts in range(4):
                eob_stats[ts].append(eob_h[blocks["tx_size"] == ts].astype(np.int64))
            luma = blocks["qtab"] == 0
            for k, (g, i) in enumerate(lb["items"]):
                sel = luma & (lb["pic_of_block"] == k)
                off = blocks["src_off"][sel].astype(np.int64) - (g * (MINIGOP + 1) + i) * pic_bytes
                r4, c4 = (off // plane_w) >> 2, (off % plane_w) >> 2
                n4 = (1 << blocks["tx_size"][sel].astype(np.int64))
                nz4 = np.zeros(((Hd + 31) // 32 * 8, (Wd + 31) // 32 * 8), bool)
                nzb = eob_h[sel] != 0
                for s in range(4):   # a block of 2^s x 2^s 4x4 units
                    m = nzb & (n4 == (1 << s))
                    for dy in range(1 << s):
                        for dx in range(1 << s):
                            nz4[r4[m] + dy, c4[m] + dx] = True
                lmi = build_lf_mode_info(B, kcell[g][i], nz4, mi_rows, mi_cols, level)
                lfm = np.zeros((sb_rows, sb_cols), dtype=B.LF_MASK_DTYPE)
                B.check(lib.svt_hip_lf_build_masks(lmi.ctypes.data_as(C.c_void_p), mi_cols, mi_rows, mi_cols, lfm.ctypes.data_as(C.c_void_p), sb_cols))
                d_lfm[g][i].copy_(torch.from_numpy(lfm.view(np.uint8).reshape(-1)))
                if g == 0:
                    with torch.no_grad():
                        resid_acc.append((d_src[0, i, :Hd].to(torch.int16) - d_pred[0, i - 1, :Hd].to(torch.int16)).abs().to(torch.float32).mean().item())
            torch.cuda.synchronize()
            with torch.cuda.stream(grp["stream"]):
                run_lf(ctx_, bt, 0)
                run_pad(ctx_, bt, 0)
            B.check(lib.svt_hip_ctx_synchronize(ctx_))
    n_blocks_step = sum(len(lb["blocks_host"]) for grp in P_main["groups"] for lb in grp["layers"])
    counts_step = [sum(lb["counts"][ts] for grp in P_main["groups"] for lb in grp["layers"]) for ts in range(4)]
    workload_stats = {"mean_abs_luma_residual_gop0": round(float(np.mean(resid_acc)), 2),
                      "mean_eob_by_tx_size": [round(float(np.concatenate(eob_stats[ts]).mean()), 1) for ts in range(4)],
                      "blocks_by_tx_size": counts_step}
    step_no = [1]   # the setup pass was step 0 (mini-GOP 0 complete in ring slot 0)
    P_single = None if (args.no_single or separate_rate or rank != 0) else build_pipeline([0], single_pairs, split="picture")
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

    def make_state():
        return {"pa_done": [None, None], "me_done": [None, None], "step": 0, "ev": []}

    def staged(S, name, stream, fn, pool):
        if name not in stages:
            return
        if pool is None:
            return fn()
        e0, e1 = pool.pair()
        e0.record(stream)
        fn()
        e1.record(stream)
        S["ev"].append((name, e0, e1))

    def step(P, S, schedule, pool=None):
        buf = S["step"] & 1
        S["step"] += 1
        ph = step_no[0] % RING          # ring slot of the newest mini-GOP of this step
        step_no[0] += 1
        # ME side (one mini-GOP ahead of the EncDec side, as the reference's ME threads are): PA writes the planes ME reads and
        # picture analysis of the NEXT mini-GOP runs beside it on its own stream
        if S["pa_done"][buf] is not None:
            me_streams[0].wait_event(S["pa_done"][buf])
        staged(S, "me", me_streams[0], lambda: run_me(P, buf, pool), pool)
        if "me" in stages:
            S["me_done"][buf] = torch.cuda.Event()
            S["me_done"][buf].record(me_streams[0])
        if "pa" in stages:
            if S["me_done"][1 - buf] is not None:
                pa_stream.wait_event(S["me_done"][1 - buf])   # ME of the previous step read the set written now
            staged(S, "pa", pa_stream, lambda: pa_call(ctx_pa, P["gops"], pa_idx, 1 - buf), pool)
            S["pa_done"][1 - buf] = torch.cuda.Event()
            S["pa_done"][1 - buf].record(pa_stream)
        # EncDec side, every stage on the group's stream in program order = dependency order
        def barrier_groups():
            """cross-stream dependency of a picture-split pipeline: the next batch of every group starts after the previous batch of all"""
            if not P["cross_sync"]:
                return
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

        if schedule == "waves":      # the five dependent temporal-layer waves of the newest mini-GOP
            for layer in range(5):
                barrier_groups()
                for grp in P["groups"]:
                    bt, lb, ctx_, st_ = grp["waves"][layer], grp["layers"][layer], grp["ctx"], grp["stream"]
                    if not bt["n"]:
                        continue
                    staged(S, "mc", st_, lambda: run_mc(ctx_, bt, ph), pool)
                    staged(S, "tq", st_, lambda: run_tq(ctx_, lb, ph), pool)
                    if separate_rate:
                        staged(S, "rate", st_, lambda: run_rate(ctx_, lb), pool)
                    staged(S, "lf", st_, lambda: run_lf(ctx_, bt, ph), pool)
                    staged(S, "pad", st_, lambda: run_pad(ctx_, bt, ph), pool)
                mark_groups()
        else:                        # diagonal: layer l of the mini-GOP l steps back -- one batch of independent pictures
            barrier_groups()
            for grp in P["groups"]:
                bt, ctx_, st_ = grp["diag"], grp["ctx"], grp["stream"]
                staged(S, "mc", st_, lambda: run_mc(ctx_, bt, ph), pool)

                def tq_all():
                    if multi_tq:
                        return run_tq_diag(ctx_, grp, ph)
                    for layer in range(5):
                        run_tq(ctx_, grp["layers"][layer], ph - layer)
                staged(S, "tq", st_, tq_all, pool)
                if separate_rate:
                    staged(S, "rate", st_, lambda: [run_rate(ctx_, lb) for lb in grp["layers"]], pool)
                staged(S, "lf", st_, lambda: run_lf(ctx_, bt, ph), pool)
                staged(S, "pad", st_, lambda: run_pad(ctx_, bt, ph), pool)
            mark_groups()
        if handoff:
            run_handoff(P, ph)

    def sync():
        for c_ in ctxs:
            B.check(lib.svt_hip_ctx_synchronize(c_))
        torch.cuda.synchronize()

    def timed_run(P, schedule, steps, warmup, barrier):
        S = make_state()
        if schedule == "diagonal":
            warmup = max(warmup, 5)   # the pipeline of five mini-GOPs has to fill
        n_ev = 2 * steps * (len(P["me_sets"][0]) + 2 + 5 * len(P["groups"]) * (5 if separate_rate else 4)) + 8
        pool = EventPool(n_ev)
        P["me_ev"] = []
        for _ in range(warmup):
            step(P, S, schedule)
        sync()
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
        stage_ms = {s: 0.0 for s in STAGES}
        for name, e0, e1 in S["ev"]:
            stage_ms[name] += e0.elapsed_time(e1) / steps
        me_launch_ms = sum(e0.elapsed_time(e1) for e0, e1 in P["me_ev"])
        return dt, t_enq / n_free, stage_ms, me_launch_ms, len(P["me_ev"])

    dt, enq_s, stage_ms, me_launch_ms, n_me_launch = timed_run(P_main, args.schedule, args.steps, args.warmup, True)
    dt = GS.reduce_elapsed(dt, dist if world > 1 else None, dev)
    if world > 1:   # nothing below needs the other ranks: they leave in step, rank 0 reports
        sync()
        dist.barrier()
        dist.destroy_process_group()
    single = {}
    if P_single is not None:
        k1 = max(4, args.steps)
        for sched in ("diagonal", "waves"):
            dt1, enq1, stage1, _, _ = timed_run(P_single, sched, k1, max(2, args.warmup), False)
            single[sched] = {"frames_per_s": MINIGOP * k1 / dt1, "ms_per_minigop": dt1 / k1 * 1e3, "steps": k1, "stage_ms": stage1, "enq": enq1}

    L = Wd * Hd
    pics_step = G * MINIGOP
    n_launch_step = len(P_main["me_sets"][0])
    stage_bytes = {
        # source luma read + padded / decimated planes written
        "pa": pics_step * int(L + (Wd + 2 * pads[0]) * (Hd + 2 * pads[0]) + (Wd // 4 + 2 * pads[2]) * (Hd // 4 + 2 * pads[2]) +
                              (((Wd // 2 + 2 * pads[1]) * (Hd // 2 + 2 * pads[1])) if l1_on else 0)),
        "me": pics_step * algorithmic_bytes_me(Wd, Hd, 2, l1_on),
        # per 8x8 unit: 96 bytes (64 luma + 2 x 16 chroma) read per reference and written once, + the 12-byte mode-info record
        "mc": int(96 * (inter_units + comp_units) + 96 * inter_units + 12 * mi_units),
        "tq": pics_step * int(7.5 * L),                # src 1.5L + pred 1.5L + qcoeff 3L + recon 1.5L (the kernel also writes dqcoeff, +3L)
        "rate": 0,
        "lf": pics_step * (3 * L + 160 * nsb),         # recon read + write (3L) + masks
        # the border of the three planes written, the edge samples read
        "pad": pics_step * int((geo.pw * geo.ph - L) + 2 * (geo.cpw * geo.cph - L // 4) + 2 * (Hd + Wd)),
    }
    kernel_of = {"pa": "svt_pa_plane_kernel", "me": "svt_me_sb_kernel", "mc": "svt_mc_kernel",
                 "tq": "svt_tq_kernel<4|8|16|32>" + ("" if separate_rate else " (+ fused coefficient rate)"),
                 "rate": "svt_rate_kernel", "lf": "svt_lf_kernel", "pad": "svt_refpad_kernel"}
    if not separate_rate:
        stages.discard("rate")
    if rank != 0:
        return
    me_ms = max(stage_ms["me"], 1e-9)
    # roofline of the dominant kernel, per launch: algorithmic bytes of a launch / its own duration, averaged over the
    # launches = (bytes of all launches) / (sum of their durations); the stream-span figure is given beside it
    per_launch_ms = me_launch_ms / max(1, n_me_launch)
    achieved = stage_bytes["me"] * args.steps / max(me_launch_ms * 1e-3, 1e-12) / 1e9 if n_me_launch else 0.0
    traffic, valu, traffic_source = None, None, None
    tj = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tj) and (Wd, Hd) == (W4K, H4K):
        rec = json.load(open(tj)).get("svt_me_sb_kernel", {})
        if rec.get("bytes_per_step"):   # the profiling run's step is one mini-GOP of one GOP
            traffic = int(rec["bytes_per_step"] * G / n_launch_step)      # per launch, like `achieved`
            traffic_source = "profiles/traffic.json (rocprofv3 --pmc passes of tools/profile_round.sh, committed; not measured in this run)"
        vi = rec.get("valu_wave_insts_per_step")
        if vi and "me" in stages:
            # the kernel's real roof: 64-lane VALU instructions issued (rocprofv3 SQ_INSTS_VALU, profiles/) per second against
            # 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz (MI355X_MICROARCH.md)
            ach = vi * G * 64 / (me_ms * 1e-3) / 1e12
            valu = {"achieved": round(ach, 2), "peak": 39.3, "unit": "T lane-ops/s", "frac": round(ach / 39.3, 4), "wave_insts_per_minigop": vi,
                    "source": "profiles/traffic.json"}
    fps = GS.aggregate_rate(pics_step, args.steps, world, dt)
    out = {
        "metric": "encoded frames/sec (block-level DSP hot path: picture analysis + ME + inter prediction from reconstructed references + "
                  "DCT/quant/recon + coefficient rate + deblock + reference padding), 4Kp60 yuv420p enc-mode 8",
        "value": round(fps, 2),
        "unit": "frames/s",
        "mpixels_per_s": round(fps * Wd * Hd / 1e6, 1),
        "single_stream_value": round(single["diagonal"]["frames_per_s"], 2) if single else None,
        "single_gop_value": round(single["waves"]["frames_per_s"], 2) if single else None,
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3),
        "ms_per_minigop": round(dt / args.steps / G * 1e3, 3),
        "host_enqueue_ms_per_step": round(enq_s * 1e3, 3),
        "setup_s": round(setup_s, 1),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u8",
        "data": "synthetic",
        "config": {"workload": f"{Wd}x{Hd} 8-bit yuv420p, -enc-mode 8 -tune 1 -q 40, 1xMI355X per rank; step = one 16-picture mini-GOP (5 temporal "
                               f"layers, B pictures, 2 reference lists) of each of {G} closed GOPs in flight; ME side on source pictures one "
                               "mini-GOP ahead; EncDec side in dependency order (" +
                               ("per step temporal layer l of the mini-GOP l steps back: every picture is predicted from reconstructions finished in "
                                "earlier steps" if args.schedule == "diagonal" else "five dependent temporal-layer waves per mini-GOP") +
                               "): inter prediction from the deblocked + padded reconstruction of the lower-layer references -> transform / quant / "
                               "recon (+ distortion + rate) -> deblocking -> reference padding, in place in the reference buffers",
                   "gops_in_flight": G, "gop_groups": n_groups, "schedule": args.schedule,
                   "stages": ["picture_analysis", "motion_estimation", "inter_prediction", "transform_quant_recon_distortion",
                              "coefficient_rate", "deblocking", "reference_padding"],
                   "stages_run": [s for s in STAGES if s in stages],
                   "pictures_per_step": pics_step, "transform_blocks_per_step": int(n_blocks_step), "q_index": Q_INDEX,
                   "deblocked_pictures": "all (as with reconstructed output enabled; the reference skips non-reference pictures otherwise)",
                   "workload_stats": workload_stats,
                   "parallelism": f"gop-shard x{world}" + (" + split-GOP reference hand-off (RCCL send/recv of the padded base-layer reconstruction)" if handoff else "")},
        "single_stream": None if not single else {
            "frames_per_s": round(single["diagonal"]["frames_per_s"], 2), "ms_per_minigop": round(single["diagonal"]["ms_per_minigop"], 3), "gops_in_flight": 1,
            "schedule": "diagonal", "steps": single["diagonal"]["steps"],
            "stage_ms_per_minigop": {s: round(single["diagonal"]["stage_ms"][s], 3) for s in STAGES if s in stages},
            "note": "ONE stream of mini-GOPs: per step temporal layer l of the mini-GOP l steps back (16 mutually independent pictures of five consecutive "
                    "mini-GOPs, each predicted from reconstructions finished in earlier steps) -- the picture-level pipelining the reference's EncDec "
                    "processes do; needs five mini-GOPs of look-ahead"},
        "single_gop": None if not single else {
            "frames_per_s": round(single["waves"]["frames_per_s"], 2), "ms_per_minigop": round(single["waves"]["ms_per_minigop"], 3), "gops_in_flight": 1,
            "schedule": "waves", "steps": single["waves"]["steps"],
            "stage_ms_per_minigop": {s: round(single["waves"]["stage_ms"][s], 3) for s in STAGES if s in stages},
            "note": "one mini-GOP at a time: its five temporal-layer waves run one after the other on one stream (only ME / picture analysis of the next "
                    "mini-GOP overlap them) -- the lowest-latency schedule, bounded by the per-picture latency of the deblocking wavefront"},
        "roofline": {"bound": "hbm", "kernel": "svt_me_sb_kernel", "achieved": round(achieved, 2), "peak": 8000.0,
                     "unit": "GB/s", "frac": round(achieved / 8000.0, 5), "traffic": traffic, "traffic_source": traffic_source,
                     "launches_per_step": n_launch_step, "avg_launch_ms": round(per_launch_ms, 4),
                     "algorithmic_bytes_per_launch": int(stage_bytes["me"] / n_launch_step),
                     "stream_span": {"kernel_ms_per_step": round(me_ms, 3), "GB_per_s": round(stage_bytes["me"] / (me_ms * 1e-3) / 1e9, 2),
                                     "frac": round(stage_bytes["me"] / (me_ms * 1e-3) / 8e12, 5),
                                     "note": f"{n_me_streams} ME streams run launches concurrently: each launch's own duration is stretched"},
                     "valu": valu},
        "kernels": {kernel_of[s]: {"ms_per_step": round(stage_ms[s], 3), "algorithmic_bytes_per_step": stage_bytes[s],
                                   "GB_per_s": round(stage_bytes[s] / (max(stage_ms[s], 1e-9) * 1e-3) / 1e9, 2),
                                   "frac_of_8TBps": round(stage_bytes[s] / (max(stage_ms[s], 1e-9) * 1e-3) / 8e12, 5)}
                    for s in STAGES if s in stages},
        "pcie_note": f"inputs are resident in HBM when the clock starts; a {Wd}x{Hd} 4:2:0 picture is {pic_bytes / 1e6:.1f} MB, so {round(fps)} frames/s "
                     f"would need {fps * pic_bytes / 1e9:.0f} GB/s of host-to-device traffic if every picture crossed PCIe (gen5 x16 sustains ~50): the "
                     "PCIe-inclusive rate of the public-API path is `api_path` (app/svt_enc_api_bench.c, DESIGN.md section 2)",
    }
    if not args.no_cpu_baseline and world == 1:   # the CPU leg runs at N = 1 only (rank 0's host cores are not shared with other ranks)
        # the CPU leg works on GOP 0 in the tight layout of round 2 (one mini-GOP, blocks grouped by size across its pictures)
        by_ts = [[] for _ in range(4)]
        for i in range(1, MINIGOP + 1):
            for ts, a in enumerate(blk_tight[0][i]):
                b = a.copy()
                for f in ("src_off", "pred_off", "recon_off"):
                    b[f] += np.uint32((i - 1) * pic_bytes)
                b["src_off"] += np.uint32(pic_bytes)
                by_ts[ts].append(b)
        blocks0 = np.concatenate([a for ts in range(4) for a in by_ts[ts]])
        pic_of0 = np.concatenate([np.full(len(a), k, np.int32) for ts in range(4) for k, a in enumerate(by_ts[ts])])
        nn0 = (16 << (2 * blocks0["tx_size"].astype(np.int64)))
        blocks0["coeff_off"] = np.concatenate([[0], np.cumsum(nn0)[:-1]]).astype(np.uint32)
        roffs, _ = T.rate_scan_offsets()
        rb0 = np.zeros(len(blocks0), dtype=B.RATE_BLOCK_DTYPE)
        rb0["coeff_off"], rb0["tx_size"], rb0["plane_type"], rb0["is_inter"] = blocks0["coeff_off"], blocks0["tx_size"], blocks0["qtab"], 1
        rb0["scan_off"] = np.array([roffs[(ts, 0)] for ts in range(4)], np.uint32)[blocks0["tx_size"]]
        rb0["ctx"] = np.random.default_rng(8).integers(0, 3, len(blocks0)).astype(np.uint8)
        lfms0 = [np.frombuffer(d_lfm[0][i].cpu().numpy(), dtype=B.LF_MASK_DTYPE).reshape(sb_rows, sb_cols) for i in range(1, MINIGOP + 1)]
        out["cpu_baseline"] = cpu_baseline(T, B, frames_all[0], src0, mi_host0, blocks0, pic_of0, qtabs, iscan, rb0, rtab, rscan, lfms0, thr, Wd, Hd, plane_w,
                                           l1_on, int(nn0.sum()))
    for c_ in ctxs:
        lib.svt_hip_ctx_destroy(c_)
    ctxs.clear()
    if world == 1 and ((Wd, Hd) == (W4K, H4K) or os.environ.get("SVT_BENCH_API_PATH")):
        torch.cuda.synchronize()
        api = api_path_rate(frames_all[0], Wd, Hd)
        if api is not None:
            out["api_path"] = api
    print(json.dumps(out))


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
            # reference padding of the deblocked pictures (in the reference the reconstruction buffer is the padded reference
            # picture; here the port's planes are copied into padded buffers first, untimed)
            padded = {}
            for i in sample:
                base = rec_h[i - 1]
                trio = []
                for pl, pd in ((base[:Hd], pad), (base[Hd:, :Wd // 2], pad // 2), (base[Hd:, Wd // 2:], pad // 2)):
                    a = np.zeros((pl.shape[0] + 2 * pd, pl.shape[1] + 2 * pd), np.uint8)
                    a[pd:pd + pl.shape[0], pd:pd + pl.shape[1]] = pl
                    trio.append(a)
                padded[i] = trio

            def pad_one(i):
                y, u, v = padded[i]
                yd = B.YuvPlanes()
                yd.y, yd.u, yd.v = y.ctypes.data + pad * y.shape[1] + pad, u.ctypes.data + (pad // 2) * u.shape[1] + pad // 2, v.ctypes.data + (pad // 2) * v.shape[1] + pad // 2
                yd.y_stride, yd.uv_stride, yd.width, yd.height = y.shape[1], u.shape[1], Wd, Hd
                assert orc.svt_oracle_ref_pad(C.byref(yd), pad, pad) == 0
            t0 = time.perf_counter()
            list(ex.map(pad_one, sample))
            t["pad"] = time.perf_counter() - t0
        return t

    t_all = run(ncpu)
    t_8 = run(8) if ncpu > 8 else t_all
    fps = lambda t: len(sample) / sum(t.values())
    return {"value": round(fps(t_all), 3), "unit": "frames/s", "cores": ncpu, "kind": "port", "cpu_model": cpu_model(),
            "value_8_cores": round(fps(t_8), 3),
            "reference_me": reference_me_rate(T, B, orc, frames, Wd, Hd, l1_on, ncpu),
            "stage_seconds_all_cores": {k: round(v, 3) for k, v in t_all.items()},
            "stage_seconds_8_cores": {k: round(v, 3) for k, v in t_8.items()},
            "sample": f"oracle (C restatement of the reference's C path, gcc -O3 -march=native on this host = auto-vectorised \"AVX2-class\" "
                      f"proxy) on {len(sample)} whole {Wd}x{Hd} pictures of the same mini-GOP (positions {sample}) through all seven stages, "
                      f"threads over independent units (ME: SB ranges, transform / rate: block ranges, prediction / deblocking / analysis: pictures); "
                      f"wall-clock with {ncpu} threads (value) and with 8 (value_8_cores)"}


if __name__ == "__main__":
    main()
