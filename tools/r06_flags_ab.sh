#!/bin/bash
# tools/r06_flags_ab.sh NAME... -- ME alone and the step with variant libraries gpurun_in/lib_NAME.so against the product
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd $ROOT
me() { env "$@" python bench.py --stages me --no-cpu-baseline --no-extras --no-single --steps 30 --warmup 6 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'], end=' ')"; }
st() { env "$@" python bench.py --no-cpu-baseline --no-extras --no-single --steps 20 --warmup 6 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'])"; }
for i in 1 2; do
  echo -n "product: ME alone "; me A=1; echo -n " step "; st A=1
  for n in "$@"; do echo -n "$n: ME alone "; me SVT_HIP_LIB=$ROOT/gpurun_in/lib_$n.so; echo -n " step "; st SVT_HIP_LIB=$ROOT/gpurun_in/lib_$n.so; done
done
