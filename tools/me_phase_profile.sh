#!/bin/bash
# tools/me_phase_profile.sh -- per-phase instruction counts of the ME kernel (profiles/r02_pmc_traffic.md, last table).
# 1. build the library with -DME_FINE_PROF (the kernel then returns after phase SVT_HIP_ME_STOP):
#      (cd svt-vp9_amd && touch csrc/me_kernel.hip && make HIPFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -DME_FINE_PROF")
# 2. run this script on the GPU box (gpurun -- 'bash tools/me_phase_profile.sh'), 3. rebuild the product library (touch + make).
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d /tmp/ph -o p -- python $ROOT/tools/me_phase_counts.py > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("/tmp/ph/**/*counter_collection.csv", recursive=True)[0]
rows = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if "svt_me_sb_kernel" not in r["Kernel_Name"]: continue
    rows.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
names = [0, 1, 19, 20, 21, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, -1]   # the stop points of tools/me_phase_counts.py, in its order
prev = None
for k, (d, v) in zip(names, rows.items()):
    w = v["SQ_WAVES"]
    cur = (v["SQ_INSTS_VALU"] / w, v["SQ_INSTS_SALU"] / w, v["SQ_INSTS_LDS"] / w)
    if prev: print("stop", k, "valu %.0f salu %.0f lds %.0f   (+%.0f +%.0f +%.0f)" % (cur + tuple(a - b for a, b in zip(cur, prev))))
    else: print("stop", k, "valu %.0f salu %.0f lds %.0f" % cur)
    prev = cur
PY
