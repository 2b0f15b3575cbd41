#!/usr/bin/env python3
"""tools/summarize_prof.py TAG [PRESET] -- turn the rocprofv3 CSVs of tools/profile_round.sh (gpurun_out/prof_TAG*) into
profiles/<round>_kernel_stats.csv, profiles/traffic.json and a markdown table on stdout.

HBM traffic follows MI355X_MICROARCH.md's HBM section: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts
64 B per 128-B request and is doubled, WRITE_SIZE is taken as reported.  The PMC passes run bench.py with one timed step;
bench.py's untimed first pass launches the same kernels once more (the transform stage without the fused rate pass), so a
stage's figure is the sum over its LAST launches_per_step dispatches.  Round 4: the EncDec side of a step is ONE call of the
picture-level driver (svt_hip_encdec_batch_device) over the 16 pictures of the diagonal batch: every kernel of the chain launches once
(the transform stage once per transform size), motion estimation once per temporal layer."""
import csv, glob, json, os, re, shutil, sys, collections

tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
preset = sys.argv[2] if len(sys.argv) > 2 else "c3"   # another BASELINE.json configuration: gpurun_out/prof_TAG_PRESET*, profiles/traffic_PRESET.json
rnd = tag[:3]
if preset != "c3":
    tag = f"{tag}_{preset}"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "gpurun_out")
prof = os.path.join(root, "profiles")
# launches of one step with ONE GOP in flight, diagonal schedule: ME per temporal layer, the transform stage per transform size (one
# launch serves the five ring slots: svt_hip_tq_rd_batch_multi_device), everything else once over the 16 pictures
# (ME launches = temporal layers of the mini-GOP: c5 has 8 pictures, 4 layers)
PER_STEP = {"svt_me_kernel": 8 if preset == "c5" else 5,   # (c5: two launches per layer -- compact layout, then the flagged SBs with the full one)
             "svt_tq_kernel": 4, "svt_lf_kernel": 1, "svt_lf_desc_kernel": 1, "svt_pa_plane_kernel": 1, "svt_mc_kernel": 1,
            "svt_refpad_kernel": 1, "svt_tq_count_kernel": 1, "svt_scan_sb_kernel": 1, "svt_scan_pic_kernel": 1, "svt_tq_emit_kernel": 1, "svt_tq_skip_kernel": 1,
            "svt_skip_update_kernel": 1, "svt_lf_mask_kernel": 1}
KERNELS = ("svt_tq_kernel", "svt_lf_kernel", "svt_lf_desc_kernel", "svt_lf_mask_kernel", "svt_pa_plane_kernel", "svt_mc_kernel", "svt_rate_kernel",
           "svt_refpad_kernel", "svt_tq_count_kernel", "svt_scan_sb_kernel", "svt_scan_pic_kernel", "svt_tq_emit_kernel", "svt_tq_skip_kernel", "svt_skip_update_kernel")


ME_NAMES = set()


def short(name):
    if re.search(r"(?<![A-Za-z_0-9])svt_tq_lane_kernel(?![A-Za-z_0-9])", name):
        return "svt_tq_kernel"  # the 4x4 instance of the TQ stage
    m = re.search(r"(?<![A-Za-z_0-9])svt_me_(fast|sb)_kernel(<[0-9]+>)?", name)
    if m:
        ME_NAMES.add(m.group(0))   # the instance that really ran: svt_me_fast_kernel<1> (csrc/me_fast.h's driver) or svt_me_sb_kernel<SPEC>
        return "svt_me_kernel"
    for k in KERNELS:
        if re.search(r"(?<![A-Za-z_0-9])" + k + r"(?![A-Za-z_0-9])", name):
            return k
    return None


def pmc(sub, prefix):
    """{kernel: {counter: sum over the last PER_STEP[kernel] dispatches}}"""
    rows = collections.defaultdict(list)
    for f in glob.glob(os.path.join(out, f"prof_{tag}_{sub}", "**", f"{prefix}_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k:
                rows[(k, r["Counter_Name"])].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    acc = collections.defaultdict(dict)
    for (k, c), v in rows.items():
        v.sort()
        acc[k][c] = sum(x for _, x in v[-PER_STEP.get(k, 1):])
    return acc


fetch, write, insts, sq = pmc("fetch", "f"), pmc("write", "w"), pmc("insts", "i"), pmc("sq", "s")
traffic = {"_comment": f"per {8 if preset == 'c5' else 16}-picture mini-GOP (one step with one GOP in flight, diagonal schedule), from rocprofv3 PMC passes on MI355X (tools/profile_round.sh, tools/summarize_prof.py): "
                       "bytes = 1024 * (2 x FETCH_SIZE + WRITE_SIZE); instruction counts = SQ_INSTS_* summed over the stage's launches of one step"}
print("| kernel | launches/step | FETCH_SIZE raw (KiB) | WRITE_SIZE (KiB) | traffic (MB) | VALU / SALU / LDS wave-instructions | waves |")
print("|---|---|---|---|---|---|---|")
for k in sorted(set(fetch) | set(write)):
    fk, wk = fetch[k].get("FETCH_SIZE", 0.0), write[k].get("WRITE_SIZE", 0.0)
    b = int(1024 * (2 * fk + wk))
    traffic[k] = {"fetch_size_kb_raw_per_step": round(fk), "write_size_kb_per_step": round(wk), "bytes_per_step": b, "launches_per_step": PER_STEP.get(k, 1)}
    i = insts.get(k, {})
    if "SQ_INSTS_VALU" in i:
        traffic[k].update(valu_wave_insts_per_step=int(i["SQ_INSTS_VALU"]), salu_wave_insts_per_step=int(i.get("SQ_INSTS_SALU", 0)),
                          lds_wave_insts_per_step=int(i.get("SQ_INSTS_LDS", 0)), waves_per_step=int(i.get("SQ_WAVES", 0)))
    print(f"| {k} | {PER_STEP.get(k, 1)} | {fk:.0f} | {wk:.0f} | {b / 1e6:.1f} | {i.get('SQ_INSTS_VALU', 0) / 1e6:.1f} M / {i.get('SQ_INSTS_SALU', 0) / 1e6:.1f} M / "
          f"{i.get('SQ_INSTS_LDS', 0) / 1e6:.1f} M | {i.get('SQ_WAVES', 0):.0f} |")
for k, v in sq.items():
    print("SQ (ME alone)", k, {c: f"{x:.4g}" for c, x in v.items()})
    if k == "svt_me_kernel":
        traffic[k]["me_alone"] = {c: int(x) for c, x in v.items()}
if "svt_me_kernel" in traffic:
    traffic["svt_me_kernel"]["rocprof_kernel_names"] = sorted(ME_NAMES)
json.dump(traffic, open(os.path.join(prof, "traffic.json" if preset == "c3" else f"traffic_{preset}.json"), "w"), indent=1)
for name in ("kernel_stats", "domain_stats"):
    src = glob.glob(os.path.join(out, f"prof_{tag}", "**", f"{tag}_{name}.csv"), recursive=True)
    if src:
        shutil.copy(src[0], os.path.join(prof, f"{rnd}_{name}.csv" if preset == "c3" else f"{rnd}_{preset}_{name}.csv"))  # profiles are named per round: r02_*
src_me = glob.glob(os.path.join(out, f"prof_{tag}_me_alone", "**", "me_kernel_stats.csv"), recursive=True)
if src_me:   # motion estimation alone on one stream (the step's pictures in one launch): the denominator of bench.py's roofline.frac
    dst = os.path.join(prof, f"{rnd}_me_alone_kernel_stats.csv" if preset == "c3" else f"{rnd}_{preset}_me_alone_kernel_stats.csv")
    shutil.copy(src_me[0], dst)
    for r in csv.DictReader(open(src_me[0])):
        if "svt_me_" in r["Name"] and "zz" not in r["Name"]:
            print(f"\nME alone: {r['Name'][:60]} calls {r['Calls']} average {float(r['AverageNs']) / 1e3:.1f} us  (-> {os.path.basename(dst)})")
src = glob.glob(os.path.join(out, f"prof_{tag}", "**", f"{tag}_kernel_stats.csv"), recursive=True)
if src:
    print("\n| kernel | calls | avg (us) | total (ms) |\n|---|---|---|---|")
    for r in csv.DictReader(open(src[0])):
        if "svt_" in r["Name"]:
            print(f"| {r['Name'][:70]} | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['TotalDurationNs']) / 1e6:.2f} |")
