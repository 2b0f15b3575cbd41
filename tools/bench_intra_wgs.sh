#!/bin/bash
run() { echo "== $*"; env "$@" python bench.py --no-cpu-baseline --no-single 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['key_frames']; print(j['value'], j['value_inter_only'], j['ms_per_step'], k['alone_ms'], k['hidden_fraction'])"; }
run SVT_HIP_INTRA_WG_PER_CU=2
run SVT_HIP_INTRA_WGS=192
run SVT_HIP_INTRA_WGS=160
run A=1
run SVT_HIP_INTRA_WGS=128
