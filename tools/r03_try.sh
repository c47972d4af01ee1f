#!/bin/bash
# quick regression + rehearsal of the dispatch work (run on the GPU box):  bash tools/r03_try.sh <tag> [knob sets...]
# (product build only)
TAG=$1; shift
mkdir -p gpurun_out/$TAG
timeout -k 10 600 python -u -m pytest tests/test_parity_gpu.py -x -q --timeout 300 --timeout-method thread -m gpu -k "bench_shape or odd_shapes or flickering or five_wave or partition or c1_full" > gpurun_out/$TAG/tests.log 2>&1 || { tail -30 gpurun_out/$TAG/tests.log; exit 1; }
tail -3 gpurun_out/$TAG/tests.log
bash tools/r03_rehearse.sh gpurun_out/$TAG c3 64 1,8 "$@" > gpurun_out/$TAG/c3.log 2>&1 || exit 1
for c in c3 c4 c5; do
  timeout -k 10 300 python bench.py --config $c --steps 4 --warmup 3 --no-cpu > gpurun_out/$TAG/bench_$c.json 2> gpurun_out/$TAG/bench_$c.err || exit 1
done
