#!/bin/bash
# tools/r06_tq_traffic.sh -- FETCH_SIZE / WRITE_SIZE of the transform stage's kernels for its launch forms (one GOP in flight, inter pictures, the
# last step's launches): SVT_HIP_TQ_SB_ORDER=0 | SVT_HIP_TQ_SB_SPLIT=0 | =1.  bytes = 1024 * (2 * FETCH_SIZE + WRITE_SIZE) (gfx950: FETCH_SIZE counts 64 B per 128-B request)
ROOT=$(cd "$(dirname "$0")/.." && pwd); OUT=$ROOT/gpurun_out; cd /tmp; export TMPDIR=/tmp
B="python $ROOT/bench.py --no-cpu-baseline --no-single --no-extras --gops 1 --groups 1 --schedule diagonal --no-key-frames --steps 1 --warmup 5"
for cfg in "SVT_HIP_TQ_SB_ORDER=0" "SVT_HIP_TQ_SB_SPLIT=0" "SVT_HIP_TQ_SB_SPLIT=1"; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/tqt; env $cfg rocprofv3 --kernel-trace --pmc $c -d /tmp/tqt -o t --output-format csv -- $B > /dev/null 2>&1
    python3 - "$cfg" $c <<'PY'
import csv, glob, sys, collections, re
f = glob.glob("/tmp/tqt/**/t_counter_collection.csv", recursive=True)[0]
rows = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "svt_tq_kernel" in n or "svt_tq_lane_kernel" in n or "svt_tq_sb_kernel" in n:
        rows[re.search(r"svt_tq_\w+(<[^>]*>)?", n).group(0)].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
tot = 0
for n, v in rows.items():
    v.sort(); tot += v[-1][1]      # the last step's launch of every instance
print(sys.argv[1], sys.argv[2], "KiB:", int(tot), {n: int(sorted(v)[-1][1]) for n, v in rows.items()})
PY
  done
done
