#!/bin/bash
# tools/r06_tq_ab.sh -- A/B of the transform stage's launch forms in the default step and with the EncDec chain alone: SVT_HIP_TQ_SB_ORDER=0 (four
# size-grouped launches), SVT_HIP_TQ_SB_SPLIT = 0 (one SB-ordered launch), 1 (4x4..16x16 | 32x32), 2 (4x4 + 8x8 | 16x16 + 32x32)
cd "$(dirname "$0")/.."
B="python bench.py --no-cpu-baseline --no-extras --no-single --steps 20 --warmup 6"
q() { python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'], {k.split('<')[0][:12]: v['ms_per_step'] for k, v in j['kernels'].items()})"; }
for rep in 1 2; do
for cfg in "SVT_HIP_TQ_SB_ORDER=0" "SVT_HIP_TQ_SB_SPLIT=0" "SVT_HIP_TQ_SB_SPLIT=1" "SVT_HIP_TQ_SB_SPLIT=2"; do echo -n "step  $cfg: "; env $cfg $B 2>/dev/null | q; done
done
for cfg in "SVT_HIP_TQ_SB_ORDER=0" "SVT_HIP_TQ_SB_SPLIT=0" "SVT_HIP_TQ_SB_SPLIT=1" "SVT_HIP_TQ_SB_SPLIT=2"; do echo -n "alone $cfg: "; env $cfg $B --gops 1 --groups 1 --no-key-frames --stages mc,lists,tq,skip,lf,pad 2>/dev/null | q; done
