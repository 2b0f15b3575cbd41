#!/bin/bash
# tools/me_phase_lds.sh [fast] -- per-phase LDS activity of the ME kernel (as tools/me_phase_profile.sh: a -DME_FINE_PROF library that returns at a
# mark; cumulative per-dispatch counters, differenced): LDS instructions, cycles the LDS is busy with them (SQ_LDS_IDX_ACTIVE) and the part of those
# that are bank-conflict cycles, per wave.  gpurun -- 'SVT_HIP_LIB=gpurun_in/lib_fineprof.so bash tools/me_phase_lds.sh fast'
ROOT=$(cd "$(dirname "$0")/.." && pwd)
if [ "$1" = fast ]; then export ME_STOPS=0,1,2,3,4,5,7,8,9,10,-1; else export ME_STOPS=0,1,2,3,4,5,6,7,8,9,10,11,12,-1; fi
case "$SVT_HIP_LIB" in ""|/*) ;; *) export SVT_HIP_LIB=$ROOT/$SVT_HIP_LIB;; esac
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/phl
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVES --output-format csv -d /tmp/phl -o p -- python $ROOT/tools/me_phase_counts.py > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections, os
f = glob.glob("/tmp/phl/**/*counter_collection.csv", recursive=True)[0]
rows = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if "svt_me_" not in r["Kernel_Name"] or "_kernel" not in r["Kernel_Name"] or "zz" in r["Kernel_Name"]: continue
    rows.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
names = [int(x) for x in os.environ["ME_STOPS"].split(",")]
prev = None
for k, (d, v) in zip(names, rows.items()):
    w = v["SQ_WAVES"]
    cur = (v["SQ_INSTS_LDS"] / w, v["SQ_LDS_IDX_ACTIVE"] / w, v["SQ_LDS_BANK_CONFLICT"] / w)
    if prev: print("stop %3d lds insts %5.0f busy %6.0f conflict %6.0f   (+%4.0f +%5.0f +%5.0f)" % ((k,) + cur + tuple(a - b for a, b in zip(cur, prev))))
    else: print("stop %3d lds insts %5.0f busy %6.0f conflict %6.0f" % ((k,) + cur))
    prev = cur
PY
