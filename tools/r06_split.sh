cd /root/repo
B="python bench.py --no-cpu-baseline --no-single --no-extras --steps 20 --warmup 6"
q() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1', 'ms_per_step', d['ms_per_step'], 'value', d['value'], {k.split('<')[0][:12]: (v['ms_per_step'], v['ms_per_step_all_streams']) for k,v in d.get('kernels',{}).items()})
"; }
$B 2>/dev/null | q all
$B --stages pa,me 2>/dev/null | q pa_me
$B --stages mc,lists,tq,skip,lf,pad 2>/dev/null | q encdec
$B --stages mc,lists,tq,skip,lf,pad --no-key-frames 2>/dev/null | q encdec_nokey
$B --no-key-frames 2>/dev/null | q all_nokey
SVT_BENCH_GROUPS=4 $B --groups 4 --no-key-frames 2>/dev/null | q all_nokey_4groups
