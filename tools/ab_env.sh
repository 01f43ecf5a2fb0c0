#!/bin/bash
# Same-box A/B of the headline bench over environment switches:  bash tools/ab_env.sh <outdir> "VAR=a" "VAR=b" ...  (each setting run twice, interleaved)
set -o pipefail
export PYTHONPATH=$PWD
OUT=$1; shift
mkdir -p $OUT
run() {
  env $1 timeout -k 10 300 python bench.py --steps ${STEPS:-40} --warmup 3 --no-cpu-baseline 2>>$OUT/err.log | tail -1 > $OUT/last.json
  python - "$1" $OUT/last.json <<'PY' | tee -a $OUT/ab.log
import json,sys
l=json.load(open(sys.argv[2]))
fam={f['family'][:28]:round(f['ms_per_step'],2) for f in l['roofline']['families']}
print(sys.argv[1], '->', round(l['ms_per_step'],2), 'ms/step', fam)
PY
}
for rep in 1 2; do for s in "$@"; do run "$s"; done; done
