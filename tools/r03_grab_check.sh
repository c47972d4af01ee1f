for rep in 1 2 3; do
for g in 1 0 5 8; do
  RT_GRAB=$g timeout -k 10 200 python bench.py --config c3 --steps 10 --warmup 3 --no-cpu > gpurun_out/g2_$g.json 2>/dev/null || exit 1
  python - gpurun_out/g2_$g.json $g <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("c3 RT_GRAB", sys.argv[2], "%.0f Mrays/s  %.3f ms/step" % (d["value"], d["ms_per_step"]), flush=True)
PY
done
done
