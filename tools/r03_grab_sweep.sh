#!/bin/bash
# units per ticket of a band's dispenser (RT_GRAB) on whole-frame launches
for rep in 1 2; do
for c in c3 c4; do
  for g in 1 2 4 8; do
    RT_GRAB=$g timeout -k 10 200 python bench.py --config $c --steps 6 --warmup 3 --no-cpu > gpurun_out/grab_${c}_$g.json 2>/dev/null || exit 1
    python - gpurun_out/grab_${c}_$g.json $c $g <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "RT_GRAB", sys.argv[3], "%.0f Mrays/s  %.3f ms/step" % (d["value"], d["ms_per_step"]), flush=True)
PY
  done
done
done
for g in 1 2 4; do
  RT_GRAB=$g timeout -k 10 300 python bench.py --config c5 --steps 2 --warmup 1 --no-cpu > gpurun_out/grab_c5_$g.json 2>/dev/null || exit 1
  python - gpurun_out/grab_c5_$g.json c5 $g <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "RT_GRAB", sys.argv[3], "%.0f Mrays/s  %.3f ms/step" % (d["value"], d["ms_per_step"]), flush=True)
PY
done
