#!/bin/bash
# tools/r06_ab.sh LIB... -- A/B of variant libraries (gpurun_in/lib_*.so) against the product in the default step: ms per step, value
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd $ROOT
run() { python bench.py --no-cpu-baseline --no-extras --no-single --steps 20 --warmup 6 $BARGS 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'], {k.split('<')[0][:12]: v['ms_per_step'] for k, v in j['kernels'].items()})"; }
echo -n "product: "; run
for l in "$@"; do echo -n "$l: "; SVT_HIP_LIB=$ROOT/gpurun_in/lib_$l.so run; done
echo -n "product: "; run
