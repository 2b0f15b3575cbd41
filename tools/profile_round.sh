#!/bin/bash
# tools/profile_round.sh TAG [PRESET] -- the rocprofv3 passes behind profiles/ (run on the GPU box, e.g. through gpurun).
# PRESET (c1 | c2 | c5; default c3): the same passes for another BASELINE.json configuration, under gpurun_out/prof_TAG_PRESET*.
# Pass 1: kernel trace + stats of the default bench command.  Passes 2-5: PMC counters, each in its own run with
# --kernel-trace only (never combined with other trace domains).  Everything lands under gpurun_out/prof_TAG*.
TAG=${1:-r05}
PRESET=${2:-c3}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --no-cpu-baseline --no-single --no-extras --preset $PRESET"
if [ "$PRESET" != c3 ]; then TAG=${TAG}_${PRESET}; fi
# the counter passes profile ONE GOP in flight (diagonal schedule): a step is then one 16-picture mini-GOP and a stage's launches of
# the last step are the last PER_STEP dispatches of its kernel (tools/summarize_prof.py)
export SVT_BENCH_ME_MERGE=0   # one ME launch per temporal layer in every pass: countable (tools/summarize_prof.py PER_STEP)
ONE="--gops 1 --groups 1 --schedule diagonal --no-key-frames"   # (inter pictures only: a key frame's launches would fall among "the last launches of the step")
rocprofv3 --kernel-trace --stats -d $OUT/prof_${TAG} -o $TAG --output-format csv -- $BENCH --steps 3 --warmup 5 > $OUT/prof_${TAG}_bench.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/prof_${TAG}_fetch -o f --output-format csv -- $BENCH $ONE --steps 1 --warmup 5 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/prof_${TAG}_write -o w --output-format csv -- $BENCH $ONE --steps 1 --warmup 5 > /dev/null 2>&1
# instruction counts of every stage (one step, all stages), and the ME kernel's issue / LDS counters with ME alone on the GPU
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES -d $OUT/prof_${TAG}_insts -o i --output-format csv -- $BENCH $ONE --steps 1 --warmup 5 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -d $OUT/prof_${TAG}_sq -o s --output-format csv -- $BENCH $ONE --steps 1 --warmup 5 --stages me > /dev/null 2>&1
# motion estimation alone on ONE stream, the step's pictures in one launch (what bench.py's me_alone pass times with HIP events): kernel trace + stats
SVT_BENCH_ME_MERGE=1 SVT_BENCH_ME_STREAMS=1 rocprofv3 --kernel-trace --stats -d $OUT/prof_${TAG}_me_alone -o me --output-format csv -- $BENCH --stages me --steps 10 --warmup 3 > /dev/null 2>&1
tail -1 $OUT/prof_${TAG}_bench.log
