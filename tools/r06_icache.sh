#!/bin/bash
# tools/r06_icache.sh -- instruction-cache counters of the ME kernels (ME alone): the 2160p M8 instance (28 KB of code) against the 1080p one (66 KB)
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd /tmp; export TMPDIR=/tmp
for p in c3 c2 c5; do
  rm -rf /tmp/ic_$p
  timeout 300 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU --output-format csv -d /tmp/ic_$p -o i -- python $ROOT/bench.py --preset $p --stages me --gops 1 --groups 1 --no-single --no-cpu-baseline --no-extras --steps 1 --warmup 2 > /dev/null 2>&1
  python3 - $p <<'PY'
import csv, glob, sys, collections
p = sys.argv[1]
f = glob.glob("/tmp/ic_%s/**/*counter_collection.csv" % p, recursive=True)
if not f: print(p, "no counters"); sys.exit()
acc = collections.defaultdict(lambda: collections.Counter())
for r in csv.DictReader(open(f[0])):
    if "svt_me_" in r["Kernel_Name"] and "kernel<" in r["Kernel_Name"]:
        acc[r["Kernel_Name"].split("(")[0][-40:]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in acc.items(): print(p, k, {a: "%.3g" % b for a, b in v.items()})
PY
done
