#!/bin/bash
# quick regression + rehearsal of the dispatch work (run on the GPU box):  bash tools/r03_try.sh <tag> [knob sets...]
TAG=$1; shift
mkdir -p gpurun_out/$TAG
timeout -k 10 600 python -u -m pytest tests/test_parity_gpu.py -x -q --timeout 300 --timeout-method thread -m gpu -k "bench_shape or odd_shapes or flickering or five_wave or partition or c1_full" > gpurun_out/$TAG/tests.log 2>&1
tail -3 gpurun_out/$TAG/tests.log
bash tools/r03_rehearse.sh gpurun_out/$TAG c3 64 1,2,4,8 "$@" > gpurun_out/$TAG/c3.log 2>&1
RT_AMD_LIB=build/stamps/librt_amd.so timeout -k 10 200 python tools/wave_timeline.py c3 8 7 > gpurun_out/$TAG/timeline_c3_w8_r7.txt 2>&1
RT_AMD_LIB=build/stamps/librt_amd.so STAMPS_RANK=7 STAMPS_WORLD=8 timeout -k 10 200 python tools/stamps_report.py c3 > gpurun_out/$TAG/stamps_c3_w8_r7.txt 2>&1
RT_AMD_LIB=build/stamps/librt_amd.so timeout -k 10 200 python tools/stamps_report.py c3 > gpurun_out/$TAG/stamps_c3.txt 2>&1
