#!/bin/bash
# tools/me_phase_profile.sh [fast] -- per-phase instruction counts AND kernel durations of the ME kernel (4K, enc-mode 8, one picture,
# two lists, temporal layer $ME_TL, default 4).  `fast`: the marks of csrc/me_fast.h's driver.
# 1. build a library with -DME_FINE_PROF (the kernel then returns at mark SVT_HIP_ME_STOP of list 0):
#      tools/build_flags_variant.sh fineprof -DME_FINE_PROF
# 2. run this script on the GPU box: gpurun -- 'SVT_HIP_LIB=gpurun_in/lib_fineprof.so bash tools/me_phase_profile.sh fast'
ROOT=$(cd "$(dirname "$0")/.." && pwd)
if [ "$1" = fast ]; then export ME_STOPS=0,1,2,3,4,5,7,8,9,10,-1; else export ME_STOPS=0,1,19,20,21,2,3,4,5,6,7,8,9,10,11,12,-1; fi
case "$SVT_HIP_LIB" in ""|/*) ;; *) export SVT_HIP_LIB=$ROOT/$SVT_HIP_LIB;; esac
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/ph /tmp/pht
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d /tmp/ph -o p -- python $ROOT/tools/me_phase_counts.py > /dev/null 2>&1
ME_PICS=${ME_PICS:-4} ME_REPS=5 rocprofv3 --kernel-trace --output-format csv -d /tmp/pht -o p -- python $ROOT/tools/me_phase_counts.py > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections, os
f = glob.glob("/tmp/ph/**/*counter_collection.csv", recursive=True)[0]
rows = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if "svt_me_" not in r["Kernel_Name"] or "_kernel" not in r["Kernel_Name"] or "zz" in r["Kernel_Name"]: continue
    rows.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
t = glob.glob("/tmp/pht/**/*kernel_trace.csv", recursive=True)[0]
durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0 for r in sorted(csv.DictReader(open(t)), key=lambda r: int(r["Start_Timestamp"])) if "svt_me_" in r["Kernel_Name"] and "zz" not in r["Kernel_Name"]]
durs = [min(durs[5 * i:5 * i + 5]) for i in range(len(durs) // 5)]   # $ME_PICS pictures per launch (default 4), best of 5
names = [int(x) for x in os.environ["ME_STOPS"].split(",")]
prev = None; pd = 0.0
for k, (d, v), us in zip(names, rows.items(), durs):
    w = v["SQ_WAVES"]
    cur = (v["SQ_INSTS_VALU"] / w, v["SQ_INSTS_SALU"] / w, v["SQ_INSTS_LDS"] / w)
    if prev: print("stop %3d valu %5.0f salu %5.0f lds %4.0f   (+%4.0f +%4.0f +%4.0f)   %7.1f us (+%6.1f)" % ((k,) + cur + tuple(a - b for a, b in zip(cur, prev)) + (us, us - pd)))
    else: print("stop %3d valu %5.0f salu %5.0f lds %4.0f   %7.1f us" % ((k,) + cur + (us,)))
    prev = cur; pd = us
PY
