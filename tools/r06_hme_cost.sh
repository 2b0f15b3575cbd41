#!/bin/bash
# tools/r06_hme_cost.sh -- what the HME levels cost at 1080p enc-mode 8: ME kernel durations (16 pictures per launch, generic instance) with levels switched off
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd /tmp; export TMPDIR=/tmp
export SVT_HIP_ME_GENERIC=1 ME_PRESET=c2 ME_PICS=16 ME_REPS=4 ME_TL=4 ME_STOPS=-1
for hack in "" "enable_hme_level_2_flag=0" "enable_hme_level_1_flag=0,enable_hme_level_2_flag=0" "enable_hme_flag=0"; do
  rm -rf /tmp/hc; ME_HACK=$hack timeout 120 rocprofv3 --kernel-trace --output-format csv -d /tmp/hc -o p -- python $ROOT/tools/me_phase_counts.py > /dev/null 2>&1
  python3 - "$hack" <<'PY'
import csv, glob, sys
t = glob.glob("/tmp/hc/**/*kernel_trace.csv", recursive=True)
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0 for r in csv.DictReader(open(t[0])) if "svt_me_sb_kernel" in r["Kernel_Name"]] if t else []
print("%-60s %s us" % (sys.argv[1] or "all three levels", " ".join("%.0f" % x for x in d)))
PY
done
