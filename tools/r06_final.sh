#!/bin/bash
# tools/r06_final.sh -- the round's bench lines: default (c3, all legs) and the other presets
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd $ROOT; mkdir -p gpurun_out
( time python bench.py > gpurun_out/r06_bench.json 2> gpurun_out/r06_bench.err ) 2> gpurun_out/r06_bench.time
for p in c1 c2 c5; do python bench.py --preset $p > gpurun_out/r06_bench_$p.json 2> gpurun_out/r06_bench_$p.err; done
tail -c 600 gpurun_out/r06_bench.json; cat gpurun_out/r06_bench.time
