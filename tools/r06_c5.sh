#!/bin/bash
# tools/r06_c5.sh -- C5 (2160p enc-mode 3 tune 0): ME parity tests, then the preset's step with and without the compact LDS layout
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd $ROOT
python -m pytest tests/test_gpu_me.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py tests/test_gpu_reference_contact.py tests/test_ref_me_process.py tests/test_gpu_golden.py -x -q -m gpu 2>&1 | tail -5
run() { python bench.py --preset c5 --no-cpu-baseline --no-extras --no-single --steps 6 --warmup 2 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'], j['roofline'].get('me_alone'))"; }
echo -n "compact: "; run
echo -n "full:    "; SVT_HIP_ME_NOCOMPACT=1 run
