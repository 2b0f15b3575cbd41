#!/usr/bin/env python3
"""Generates tests/golden/*.npz by running the REFERENCE (oracle/_ref, built from /root/reference by oracle/Makefile)
on seeded inputs.  Run in the build container only:  python tests/gen_golden.py
The fixtures hold inputs' seeds/parameters and the reference's outputs (data only)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import me_configs as MC  # noqa: E402
import svt_testlib as T  # noqa: E402

G = T.GOLDEN_DIR
QUANT_GOLDEN_DELTAS = ((0, 0, 0), (-3, 4, -5))
# (width, height, frame_rate_q16, numerator, denominator, [(byte_count, pts), ...]) -> the reference application's IVF headers
IVF_GOLDEN_CASES = ((3840, 2160, 60 << 16, 0, 0, ((1000, 5), (7, (1 << 32) + 5))), (640, 360, 30 << 16, 30000, 1001, ((0, 0), (4294967295, (1 << 63) + 12345))),
                    (8192, 4320, (24 << 16) + 4660, 0, 1, ((1, 1),)), (64, 64, 0, 25, 1, ()))
RATE_GOLDEN_CASES = ((1, 256, 128, False), (2, 128, 64, False), (11, 128, 64, True))
MC_GOLDEN_CASES = ((1, 192, 128, 1), (2, 128, 72, 1), (3, 64, 64, 0), (4, 200, 136, 1))


def gen_scan():
    np.savez_compressed(os.path.join(G, "vp9_scan_tables.npz"), **T.ref_scan_tables())


def gen_me():
    # ---- ME: reference motion_estimate_sb results ----
    me = {}
    for name in MC.PRESETS:
        for nl, tl, clip in ((2, 1, "subpel"), (1, 0, "int"), (2, 3, "subpel")):
            gen = T.gen_clip if clip == "int" else T.gen_clip_subpel
            pics = [T.PaPic(f) for f in gen(264, 200, 3, 11)]
            res, rc = T.ref_me_picture(pics[1], pics[0], pics[2] if nl == 2 else None, MC.preset(name, nl, tl))
            me[f"{name}|{nl}|{tl}|{clip}"] = res
    np.savez_compressed(os.path.join(G, "me_reference.npz"), **me)


def gen_tq():
    # ---- TQ: reference kernels on the seeded block list of make_tq_case ----
    tq = {}
    for seed in (1, 2):
        case = T.make_tq_case(seed)
        n_blocks = len(case["blocks"])
        q_all = np.zeros(case["n_coeff"], np.int16)
        dq_all = np.zeros(case["n_coeff"], np.int16)
        eobs = np.zeros(n_blocks, np.uint16)
        recon = np.zeros_like(case["src"])
        for i, b in enumerate(case["blocks"]):
            ts, tt = int(b["tx_size"]), int(b["tx_type"]) if b["tx_size"] < 3 else 0
            n = T.TX_N[ts]
            y, x = divmod(int(b["src_off"]), case["src"].shape[1])
            res = case["src"][y:y + n, x:x + n].astype(np.int16) - case["pred"][y:y + n, x:x + n].astype(np.int16)
            co = T.ref_fwd_txfm(res, ts, tt, bool(b["partial32"]))
            q, dq, eob = T.ref_quantize(co, ts, tt, case["qtabs"][int(b["qtab"])])
            off = int(b["coeff_off"])
            q_all[off:off + n * n], dq_all[off:off + n * n], eobs[i] = q, dq, eob
            pred = case["pred"][y:y + n, x:x + n]
            recon[y:y + n, x:x + n] = T.ref_inv_add(dq, pred, ts, tt, eob) if eob else pred
        tq[f"q{seed}"], tq[f"dq{seed}"], tq[f"eob{seed}"], tq[f"recon{seed}"] = q_all, dq_all, eobs, recon
    np.savez_compressed(os.path.join(G, "tq_reference.npz"), **tq)


def gen_lf():
    # ---- LF: reference eb_vp9_loop_filter_frame ----
    lf = {}
    for (w, h, seed, sharp) in ((200, 136, 2, 0), (328, 200, 3, 4), (72, 72, 5, 0)):
        y, u, v = T.ref_lf_frame(T.make_lf_case(seed, w, h, sharp))
        lf[f"y|{w}|{h}|{seed}|{sharp}"], lf[f"u|{w}|{h}|{seed}|{sharp}"], lf[f"v|{w}|{h}|{seed}|{sharp}"] = y, u, v
    np.savez_compressed(os.path.join(G, "lf_reference.npz"), **lf)


def gen_lf_masks():
    # ---- LF masks: reference eb_vp9_setup_mask on random mode-info grids (same cases as tests/test_lf_masks.py) ----
    lm = {}
    for (seed, mi_rows, mi_cols) in ((1, 8, 8), (2, 27, 41), (3, 17, 9), (4, 5, 3), (5, 34, 60)):
        cells, lvl, _ = T.gen_mode_info_grid(seed, mi_rows, mi_cols)
        r = T.ref_lf_build_masks(cells, lvl, mi_rows, mi_cols)
        for n in r.dtype.names:
            lm[f"lfm_{seed}_{n}"] = np.ascontiguousarray(r[n])
    np.savez_compressed(os.path.join(G, "lf_masks_reference.npz"), **lm)


def gen_mc():
    # ---- inter prediction: the reference's inter_prediction() on seeded mode-info grids (tests/test_mc.py) ----
    assert T.have_ref("ref_mc_frame")
    mc = {}
    for (seed, w, h, sub) in MC_GOLDEN_CASES:
        case = T.make_mc_case(seed, width=w, height=h, use_subpel=sub)
        y, u, v = T.ref_mc_frame(case)
        mc[f"y|{seed}|{w}|{h}|{sub}"], mc[f"u|{seed}|{w}|{h}|{sub}"], mc[f"v|{seed}|{w}|{h}|{sub}"] = y, u, v
    np.savez_compressed(os.path.join(G, "mc_reference.npz"), **mc)


def gen_rate():
    # ---- coefficient rate estimation: the reference's tables (token costs of the default coefficient probabilities, value
    # / cat6 cost tables, scan orders with neighbours) and its coeff_rate_estimate() on the blocks of tests/test_rate.py ----
    assert T.have_ref("ref_rate_blocks")
    dummy = dict(qcoeff=np.zeros(16, np.int16), blocks=np.zeros(1, dtype=T.B.RATE_BLOCK_DTYPE), tx_type=np.zeros(1, np.int32))
    _, tab, scan = T.ref_rate_run(dummy)
    rate = {k: np.ascontiguousarray(tab[k]) for k in ("token_costs", "value_cost", "cat6_low_cost", "cat6_high_cost")}
    rate["scan"] = scan
    for (seed, w, h, ext) in RATE_GOLDEN_CASES:
        case = T.make_rate_case(seed, width=w, height=h, scan=scan, extreme=ext)
        rate[f"bits|{seed}|{w}|{h}|{int(ext)}"] = T.ref_rate_run(case)[0]
    np.savez_compressed(os.path.join(G, "rate_reference.npz"), **rate)


def gen_quant():
    # ---- quantiser tables: the reference's eb_vp9_init_quantizer for all 256 q indices, two delta settings ----
    import subprocess
    qt = {}
    for deltas in QUANT_GOLDEN_DELTAS:
        out = subprocess.check_output([os.path.join(T.REF_DIR, "ref_quant_tables")] + [str(d) for d in deltas]).decode()
        qt["|".join(map(str, deltas))] = np.array([[int(x) for x in line.split()] for line in out.strip().splitlines()], np.int32)
    np.savez_compressed(os.path.join(G, "quant_reference.npz"), **qt)


def gen_qp_scaling():
    # ---- fixed-QP layer scaling: the reference's eb_vp9_compute_qdelta + delta_rate tables (oracle/_ref/ref_qp_scaling) ----
    import subprocess
    out = subprocess.check_output([os.path.join(T.REF_DIR, "ref_qp_scaling")]).decode()
    rows = np.array([[int(x) for x in line.split()] for line in out.strip().splitlines()], np.int16)   # tune, levels, layer, qp, base_qindex
    np.savez_compressed(os.path.join(G, "qp_scaling_reference.npz"), rows=rows)


def gen_ivf():
    # ---- IVF container headers: the reference application's write_ivf_stream_header / write_ivf_frame_header ----
    ivf = {}
    for k, (w, h, fr, num, den, frames) in enumerate(IVF_GOLDEN_CASES):
        ivf[str(k)] = np.frombuffer(T.ref_ivf_headers(w, h, fr, num, den, frames), np.uint8)
    np.savez_compressed(os.path.join(G, "ivf_reference.npz"), **ivf)


def gen_sad_loop():
    # ---- stand-alone exhaustive SAD search: the reference's eb_vp9_sad_loop_kernel on the jobs of make_sad_loop_case ----
    np.savez_compressed(os.path.join(G, "sad_loop_reference.npz"), **{str(seed): T.ref_sad_loop_case(T.make_sad_loop_case(seed)) for seed in (1, 2)})


def gen_sb_stats():
    # ---- stationary-edge flags: the reference's part1 / part2 + eb_vp9_sb_params_init on the cases of tests/test_me_side.py ----
    np.savez_compressed(os.path.join(G, "sb_stats_reference.npz"), **{str(c[0]): T.ref_me_stationary_edge(T.make_sb_stats_case(*c)) for c in T.SB_STATS_CASES})


def gen_api():
    # ---- public encoder API: struct layouts, library defaults, level tables, parameter-check verdicts of the reference ----
    np.savez_compressed(os.path.join(G, "api_reference.npz"), **T.ref_api())


def gen_lf_params():
    # ---- LF parameters: the reference's eb_vp9_loop_filter_init (sharpness 0..7) and eb_vp9_pick_filter_level (all q) ----
    np.savez_compressed(os.path.join(G, "lf_params_reference.npz"), **T.ref_lf_params())


def gen_me_presets():
    # ---- ME presets: the reference's eb_vp9_signal_derivation_me_kernel_{oq,sq} for every picture class / mode / tune ----
    np.savez_compressed(os.path.join(G, "me_presets_reference.npz"), **T.ref_me_presets())


def gen_refpad():
    # ---- recon -> reference padding: the reference's pad_ref_and_set_flags on seeded buffers (tests/test_refpad.py) ----
    assert T.have_ref("ref_refpad")
    out = {}
    for args in T.REFPAD_GOLDEN_CASES:
        for k, b in enumerate(T.ref_ref_pad(T.make_refpad_case(*args))):
            out[f"{args[0]}|{k}"] = b
    np.savez_compressed(os.path.join(G, "refpad_reference.npz"), **out)


def gen_avg_ssd():
    # ---- eb_vp9_combined_averaging_ssd: the one leaf of the SSD fractional search that runs here (no Log2f) ----
    np.savez_compressed(os.path.join(G, "avg_ssd_reference.npz"), **{str(seed): T.ref_avg_ssd_jobs(T.make_avg_ssd_jobs(seed)) for seed in (1, 2)})


def gen_pd_split():
    # ---- mini-GOP window split: the reference's picture-decision functions for pre-assignment buffers of 2..16 pictures ----
    np.savez_compressed(os.path.join(G, "pd_split_reference.npz"), split=T.ref_minigop_split())


def gen_encdec_flags():
    # ---- the EncDec kernel's stage flags for every (tune, enc-mode, temporal layer, reference / not) from the reference's own derivation ----
    assert T.have_ref("ref_refpad")
    np.savez_compressed(os.path.join(G, "encdec_flags_reference.npz"), flags=T.ref_encdec_flags())


SECTIONS = ("scan", "me", "tq", "lf", "lf_masks", "mc", "rate", "quant", "ivf", "lf_params", "me_presets", "sad_loop", "sb_stats", "api", "refpad", "avg_ssd", "pd_split", "encdec_flags", "qp_scaling")


def main():
    """python tests/gen_golden.py [section ...]  (no argument: every section)"""
    assert T.ref_kernels() is not None and T.have_ref("ref_me_sb") and T.have_ref("ref_lf_frame"), "build oracle/_ref first"
    for n in sys.argv[1:] or SECTIONS:
        globals()["gen_" + n]()
    for f in sorted(os.listdir(G)):
        print(f, os.path.getsize(os.path.join(G, f)))


if __name__ == "__main__":
    main()
