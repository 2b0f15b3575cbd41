#!/bin/bash
# tools/r06_baseline.sh TAG -- round-6 measurement set on the GPU box: kernel traces of ME alone (one stream, the step's 64 pictures in one
# launch) and of the EncDec chain alone (one GOP, one stream, inter pictures), then the default bench line.
TAG=${1:-r06}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
B="python $ROOT/bench.py --no-cpu-baseline --no-single --no-extras"
SVT_BENCH_ME_STREAMS=1 rocprofv3 --kernel-trace --stats -d $OUT/prof_${TAG}_me_alone -o me --output-format csv -- $B --stages me --steps 10 --warmup 3 > $OUT/prof_${TAG}_me_alone.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/prof_${TAG}_ed_alone -o ed --output-format csv -- $B --gops 1 --groups 1 --no-key-frames --stages mc,lists,tq,skip,lf,pad --steps 10 --warmup 6 > $OUT/prof_${TAG}_ed_alone.log 2>&1
for t in me_alone ed_alone; do f=$(find $OUT/prof_${TAG}_$t -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/${TAG}_${t}_kernel_stats.csv && cut -d, -f1-4 "$f" | head -24; done
cd $ROOT && python bench.py ${BENCH_ARGS:---no-cpu-baseline} > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; tail -c 600 $OUT/${TAG}_bench.json
