#!/bin/bash
# bench value vs the number of streams / hardware queues
run() { echo "== $*"; env "$@" python bench.py --no-cpu-baseline --no-single --no-extras 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j[\"value\"], j[\"ms_per_step\"])"; }
run A=1
run SVT_BENCH_KEY_STREAMS=1
run SVT_BENCH_ME_STREAMS=1
run SVT_BENCH_ME_STREAMS=1 SVT_BENCH_KEY_STREAMS=1
run GPU_MAX_HW_QUEUES=8
run GPU_MAX_HW_QUEUES=2
run A=1
