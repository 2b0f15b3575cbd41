#!/bin/bash
cd $GRAFT_REPO_ROOT
python tools/me_fuzz.py 80 9 fast 2>&1 | tail -1
python -m pytest tests/test_gpu_me.py tests/test_gpu_fullsize.py tests/test_gpu_reference_contact.py -x -q -m gpu 2>&1 | tail -1
run() { env "$@" python bench.py --stages me --no-cpu-baseline --no-extras --no-single --steps 30 --warmup 6 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'])"; }
for i in 1 2 3; do echo -n "base: "; run SVT_HIP_LIB=$GRAFT_REPO_ROOT/gpurun_in/lib_base.so; echo -n "new:  "; run A=1; done
runs() { env "$@" python bench.py --no-cpu-baseline --no-extras --no-single --steps 20 --warmup 6 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'])"; }
for i in 1 2; do echo -n "step base: "; runs SVT_HIP_LIB=$GRAFT_REPO_ROOT/gpurun_in/lib_base.so; echo -n "step new:  "; runs A=1; done
