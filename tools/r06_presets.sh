#!/bin/bash
# tools/r06_presets.sh [ENV=VAL ...] -- ME parity tests, then value / ms per step / ME alone of the c1, c2, c5 (and c3) presets
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd $ROOT
python -m pytest tests/test_gpu_me.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py tests/test_gpu_reference_contact.py tests/test_ref_me_process.py tests/test_gpu_golden.py -x -q -m gpu 2>&1 | tail -3
for p in c2 c1 c5 c3; do echo -n "$p: "; env "$@" python bench.py --preset $p --no-cpu-baseline --no-extras --no-single --steps 8 --warmup 3 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'], j['roofline'].get('me_alone'))"; done
