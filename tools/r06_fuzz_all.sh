#!/bin/bash
# tools/r06_fuzz_all.sh [seed] -- the randomised parity sweeps of every stage against the oracle (GPU), fresh seed, each under a timeout
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd $ROOT; S=${1:-606}
for job in "me_fuzz.py 200 $S" "me_fuzz.py 100 $S fast" "me_fuzz.py 150 $S c5" "tq_fuzz.py 150 $S" "lf_fuzz.py 150 $S" "mc_fuzz.py 150 $S" "intra_fuzz.py 300 $S" "encdec_fuzz.py 100 $S"; do
  echo -n "$job: "; timeout 900 python tools/$job 2>&1 | tail -1
done
for so in 0 1; do echo -n "encdec_fuzz.py 60 $S (SVT_HIP_TQ_SB_ORDER=$so): "; SVT_HIP_TQ_SB_ORDER=$so timeout 600 python tools/encdec_fuzz.py 60 $((S + 1)) 2>&1 | tail -1; done
