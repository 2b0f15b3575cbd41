#!/bin/bash
# tools/r06_api_trace.sh [recon 0|1] [pictures] -- kernel + memory-copy trace of the public-API path (app/svt_enc_api_bench): who is busy when
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd /tmp; export TMPDIR=/tmp
RECON=${1:-1}; N=${2:-300}
python3 - <<PY
import sys, os, numpy as np
sys.path.insert(0, "$ROOT/tests")
import svt_testlib as T
W, H = 3840, 2160
with open("/tmp/clip_tr.yuv", "wb") as f:
    for y in T.gen_clip(W, H, 17, 5):
        y = np.ascontiguousarray(y); f.write(y.tobytes()); f.write((y[::2, ::2] // 2 + 32).astype(np.uint8).tobytes()); f.write(np.full((H // 2, W // 2), 128, np.uint8).tobytes())
PY
rm -rf /tmp/apitr
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/apitr -o t -- $ROOT/app/svt_enc_api_bench /tmp/clip_tr.yuv 3840 2160 17 $N 8 1 $RECON 2>&1 | tail -1
python3 - <<'PY'
import csv, glob, collections
kt = glob.glob("/tmp/apitr/**/*kernel_trace.csv", recursive=True)[0]
mt = glob.glob("/tmp/apitr/**/*memory_copy_trace.csv", recursive=True)
ev = []
for r in csv.DictReader(open(kt)):
    n = r["Kernel_Name"]
    fam = "me" if "svt_me_" in n else "lf" if "svt_lf_kernel" in n else "tq" if "svt_tq_kernel" in n or "svt_tq_lane" in n else "mc" if "svt_mc_" in n else "intra" if "svt_intra" in n else "pa" if "svt_pa_" in n else "copy2d" if "copyBuffer" in n or "copy" in n.lower() else "small"
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), fam))
if mt:
    for r in csv.DictReader(open(mt[0])):
        d = r.get("Direction", "")
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "H2D" if "HOST_TO_DEVICE" in d else "D2H" if "DEVICE_TO_HOST" in d else "D2D"))
ev.sort()
t0, t1 = ev[0][0], max(e[1] for e in ev)
a, b = t0 + (t1 - t0) * 0.3, t0 + (t1 - t0) * 0.9
pts = []
for s, e, f in ev:
    s, e = max(s, a), min(e, b)
    if e > s: pts.append((s, 1, f)); pts.append((e, -1, f))
pts.sort()
cnt = collections.Counter(); busy = collections.Counter(); last = a; nothing = 0; nokern = 0
for t, d, f in pts:
    dt = t - last
    act = [k for k, v in cnt.items() if v > 0]
    if dt > 0:
        for k in act: busy[k] += dt
        if not act: nothing += dt
        if not [k for k in act if k not in ("H2D", "D2H", "D2D")]: nokern += dt
    cnt[f] += d; last = t
tot = b - a
print("window %.1f ms; nothing in flight %.1f %%; no kernel in flight %.1f %%" % (tot / 1e6, 100 * nothing / tot, 100 * nokern / tot))
for k in sorted(busy, key=lambda k: -busy[k]): print("  %-7s in flight %5.1f %%" % (k, 100 * busy[k] / tot))
h2d = [(s, e) for s, e, f in ev if f == "H2D" and s >= a and e <= b and e - s > 50000]
d2h = [(s, e) for s, e, f in ev if f == "D2H" and s >= a and e <= b and e - s > 50000]
for nm, L in (("H2D", h2d), ("D2H", d2h)):
    if L: print("  %s: %d large copies, average %.1f us (12.4 MB -> %.1f GB/s)" % (nm, len(L), sum(e - s for s, e in L) / len(L) / 1e3, 12.44e6 / (sum(e - s for s, e in L) / len(L))))
PY
