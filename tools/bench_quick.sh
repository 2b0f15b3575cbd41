# tools/bench_quick.sh [bench.py args] -- one short line per run: step time, host enqueue time, per-stage kernel spans
# usage: bq.sh [bench args]  -> prints top-level ms_per_step and kernel ms
python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c '
import sys, json
for l in sys.stdin:
    if l.startswith("{"):
        d = json.loads(l); print("ms_per_step", d["ms_per_step"], "host", d.get("host_enqueue_ms_per_step"), {k.split("<")[0][:14]: v["ms_per_step"] for k, v in d.get("kernels", {}).items()})
'
