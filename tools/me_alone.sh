#!/bin/bash
# tools/me_alone.sh [lib.so ...] -- ME alone on the GPU (bench.py --stages me, one GOP in flight): ms per mini-GOP for the product library
# and for every variant library given (gpurun_in/lib_*.so built by tools/build_*_variant.sh).
ROOT=$(cd "$(dirname "$0")/.." && pwd)
run() { python $ROOT/bench.py --stages me --gops 1 --groups 1 --no-single --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_minigop'],3), 'ms/mini-GOP', d['roofline']['avg_launch_ms'], 'ms/launch')"; }
echo -n "product: "; run
for l in "$@"; do echo -n "$l: "; SVT_HIP_LIB=$ROOT/$l run; done
