#!/usr/bin/env python3
"""tools/summarize_prof.py TAG -- turn the rocprofv3 CSVs of tools/profile_round.sh (gpurun_out/prof_TAG*) into
profiles/TAG_kernel_stats.csv, profiles/traffic.json and a markdown table on stdout.

HBM traffic follows MI355X_MICROARCH.md's HBM section: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts
64 B per 128-B request and is doubled, WRITE_SIZE is taken as reported."""
import csv, glob, json, os, re, shutil, sys, collections

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "gpurun_out")
prof = os.path.join(root, "profiles")


def short(name):
    if re.search(r"(?<![A-Za-z_0-9])svt_tq_lane_kernel(?![A-Za-z_0-9])", name):
        return "svt_tq_kernel"  # the 4x4 instance of the TQ stage
    for k in ("svt_me_sb_kernel", "svt_tq_kernel", "svt_lf_kernel", "svt_lf_desc_kernel", "svt_pa_plane_kernel", "svt_pa_meanvar_kernel", "svt_me_zz_sad_kernel", "svt_mc_kernel", "svt_rate_kernel"):
        if re.search(r"(?<![A-Za-z_0-9])" + k + r"(?![A-Za-z_0-9])", name):
            return k
    return None


def pmc(sub, prefix):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.Counter()
    for f in glob.glob(os.path.join(out, f"prof_{tag}_{sub}", "**", f"{prefix}_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k:
                acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
                n[(k, r["Counter_Name"])] += 1
    return acc, n


fetch, nf = pmc("fetch", "f")
write, nw = pmc("write", "w")
sq, ns = pmc("sq", "s")
REPS = {"svt_pa_plane_kernel": 4, "svt_mc_kernel": 4, "svt_rate_kernel": 4}  # bench.py times these stages in 4 repetitions of their own, once per run
traffic = {"_comment": "HBM traffic per 16-picture step from rocprofv3 PMC passes on MI355X (tools/profile_round.sh, "
                       "tools/summarize_prof.py): bytes = 1024 * (2 x FETCH_SIZE + WRITE_SIZE)"}
print("| kernel | launches/step | FETCH_SIZE raw (KiB) | WRITE_SIZE (KiB) | traffic (MB) |")
print("|---|---|---|---|---|")
for k in sorted(set(fetch) | set(write)):
    rep = REPS.get(k, 1)
    fk, wk = fetch[k].get("FETCH_SIZE", 0.0) / rep, write[k].get("WRITE_SIZE", 0.0) / rep
    b = int(1024 * (2 * fk + wk))
    traffic[k] = {"fetch_size_kb_raw_per_step": round(fk), "write_size_kb_per_step": round(wk), "bytes_per_step": b,
                  "launches_per_step": nf[(k, "FETCH_SIZE")] // rep}
    print(f"| {k} | {nf[(k, 'FETCH_SIZE')] // rep} | {fk:.0f} | {wk:.0f} | {b / 1e6:.1f} |")
for k, v in sq.items():  # instruction counts of the SQ pass (ME stage alone): per step for ME
    if k in traffic and "SQ_INSTS_VALU" in v:
        traffic[k]["valu_wave_insts_per_step"] = int(v["SQ_INSTS_VALU"] / REPS.get(k, 1))
json.dump(traffic, open(os.path.join(prof, "traffic.json"), "w"), indent=1)
for k, v in sq.items():
    print("SQ", k, {c: f"{x:.4g}" for c, x in v.items()})
for name in ("kernel_stats", "domain_stats"):
    src = glob.glob(os.path.join(out, f"prof_{tag}", "**", f"{tag}_{name}.csv"), recursive=True)
    if src:
        shutil.copy(src[0], os.path.join(prof, f"{tag[:3]}_{name}.csv"))  # profiles are named per round: r01_*
src = glob.glob(os.path.join(out, f"prof_{tag}", "**", f"{tag}_kernel_stats.csv"), recursive=True)
if src:
    print("\n| kernel | calls | avg (us) | total (ms) |\n|---|---|---|---|")
    for r in csv.DictReader(open(src[0])):
        if short(r["Name"]):
            print(f"| {r['Name'][:60]} | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['TotalDurationNs']) / 1e6:.2f} |")
