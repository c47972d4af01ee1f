#!/bin/bash
# A/B of environment knobs on the GPU box:  bash tools/ab.sh <out-dir> <config> <bench args> -- VAR=val[,VAR=val] ...
# Runs bench.py once per knob set (interleaved, twice) and prints value / ms_per_step per run.
O=$1; shift; CFG=$1; shift
ARGS=()
while [ "$1" != "--" ] && [ $# -gt 0 ]; do ARGS+=("$1"); shift; done
shift
mkdir -p $O
for rep in 1 2; do
  for knobs in "$@"; do
    tag=$(echo "$knobs" | tr ',=/' '___')
    env $(echo "$knobs" | tr ',' ' ') timeout -k 10 300 python bench.py --config $CFG --no-cpu "${ARGS[@]}" > $O/${CFG}_${tag}_$rep.json 2> $O/${CFG}_${tag}_$rep.err || echo "FAILED $knobs"
    python3 - "$O/${CFG}_${tag}_$rep.json" "$knobs" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    print("%-40s %10.1f Mrays/s %9.3f ms/step  kernel %s %.3f ms" % (sys.argv[2], d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["avg_launch_ms"]))
except Exception as e:
    print(sys.argv[2], "no result", e)
PY
  done
done
