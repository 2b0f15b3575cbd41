#!/usr/bin/env python3
"""tools/timeline.py KERNEL_TRACE.csv [T0_FRACTION T1_FRACTION] -- concurrency picture of a rocprofv3 --kernel-trace run of bench.py: per
kernel family the union of its busy intervals, the time during which it is the ONLY family on the GPU, and the pairwise overlaps, over the
middle of the trace (default 40 % .. 90 %: the timed steps)."""
import csv, sys, collections
f = sys.argv[1]
lo, hi = (float(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (0.4, 0.9)
FAM = (("me", "svt_me_"), ("lf", "svt_lf_kernel"), ("tq", "svt_tq_kernel"), ("tq", "svt_tq_lane"), ("mc", "svt_mc_"), ("intra", "svt_intra_"), ("pa", "svt_pa_"))
rows = []
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    fam = next((a for a, b in FAM if b in n), "small" if "svt_" in n else None)
    if fam:
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), fam))
rows.sort()
t_min, t_max = rows[0][0], max(r[1] for r in rows)
a, b = t_min + (t_max - t_min) * lo, t_min + (t_max - t_min) * hi
ev = []
for s, e, fam in rows:
    s, e = max(s, a), min(e, b)
    if e > s:
        ev.append((s, 1, fam)); ev.append((e, -1, fam))
ev.sort()
cnt = collections.Counter()
excl, busy, pair = collections.Counter(), collections.Counter(), collections.Counter()
idle, last = 0, a
for t, d, fam in ev:
    dt = t - last
    act = sorted(k for k, v in cnt.items() if v > 0)
    if dt > 0:
        if not act: idle += dt
        for k in act: busy[k] += dt
        if len(act) == 1: excl[act[0]] += dt
        pair["+".join(act)] += dt
    cnt[fam] += d
    last = t
tot = b - a
print(f"window {tot / 1e6:.2f} ms; idle {100 * idle / tot:.1f} %")
for k in sorted(busy, key=lambda k: -busy[k]):
    print(f"  {k:6s} busy {100 * busy[k] / tot:5.1f} %   alone on the GPU {100 * excl[k] / tot:5.1f} %")
print("combinations (share of the window):")
for k, v in pair.most_common(12):
    print(f"  {k or '(idle)':28s} {100 * v / tot:5.1f} %")
