#!/bin/bash
# tools/prof_stages.sh TAG STAGES [ENV...] -- kernel trace (+stats) of bench.py restricted to some stages (stand-alone kernel times)
TAG=$1; STAGES=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
env "$@" rocprofv3 --kernel-trace --stats -d $OUT/prof_${TAG} -o $TAG --output-format csv -- python $ROOT/bench.py --no-cpu-baseline --steps 5 --warmup 2 --stages $STAGES > $OUT/prof_${TAG}.log 2>&1
f=$(find $OUT/prof_${TAG} -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cut -d, -f1-4,6 "$f" | head -14
