#!/bin/bash
# tools/stage_alone.sh STAGES [lib.so ...] -- the named stages alone on the GPU (bench.py --stages, one GOP in flight, diagonal): ms per
# mini-GOP for the product library and for every variant library given; then the default bench for each (value).
ROOT=$(cd "$(dirname "$0")/.." && pwd)
ST=$1; shift
alone() { python $ROOT/bench.py --stages $ST --gops 1 --groups 1 --no-single --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_minigop'],3), 'ms/mini-GOP alone', {k.split('_')[1]:v['ms_per_step'] for k,v in d['kernels'].items()})"; }
full() { python $ROOT/bench.py --no-single --no-cpu-baseline --steps 12 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   full bench', d['value'], 'frames/s', d['ms_per_minigop'])"; }
echo -n "product: "; alone; full
for l in "$@"; do echo -n "$l: "; export SVT_HIP_LIB=$ROOT/$l; alone; full; done
