#!/bin/bash
# A/B of two builds of the library on the GPU box, interleaved:  bash tools/r03_ab.sh <tag> <libA> <libB> [config ...]
# ("-" = the product library); bench.py per config (full spp) and the eighth-partition rehearsal of c3
TAG=$1; A=$2; B=$3; shift 3
CONFIGS=${@:-c3 c4 c5}
O=gpurun_out/$TAG; mkdir -p $O
for rep in 1 2; do
  for c in $CONFIGS; do
    for v in A B; do
      lib=$A; [ $v = B ] && lib=$B
      if [ "$lib" = "-" ]; then unset RT_AMD_LIB; else export RT_AMD_LIB=$lib; fi
      timeout -k 10 300 python bench.py --config $c --steps 4 --warmup 3 --no-cpu > $O/bench_${c}_${v}_$rep.json 2> $O/bench_${c}_${v}_$rep.err || exit 1
      python - $O/bench_${c}_${v}_$rep.json $c $v <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], sys.argv[3], "%.0f Mrays/s  %.3f ms/step" % (d["value"], d["ms_per_step"]), flush=True)
PY
    done
  done
  for v in A B; do
    lib=$A; [ $v = B ] && lib=$B
    if [ "$lib" = "-" ]; then unset RT_AMD_LIB; else export RT_AMD_LIB=$lib; fi
    echo "rehearsal $v"; REHEARSAL_WORLDS=1,8 timeout -k 10 300 python tools/partition_rehearsal.py c3 64 10 || exit 1
  done
done
