#!/bin/bash
run() { echo "== $*"; env "$@" python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j[\"value\"], j[\"ms_per_step\"], j[\"single_gop_value\"], j[\"single_stream_value\"])"; }
run A=1
run SVT_HIP_LF_ROWS=34 SVT_HIP_LF_EARLY=1
run SVT_HIP_LF_ROWS=34
run SVT_HIP_LF_ROWS=24 SVT_HIP_LF_EARLY=1
run A=1
