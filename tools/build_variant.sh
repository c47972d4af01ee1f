#!/bin/bash
# Diagnostic builds of the library: bash tools/build_variant.sh <name> <extra hipcc flags>
#   stamps  -DRT_STAMPS       in-kernel s_memtime stamps (tools/stamps_report.py)
#   stats   -DRT_TRAV_STATS   traversal counters (tools/trav_stats.py)
#   packed  -DRT_PACKED -fslp-vectorize   the ray pairs' arithmetic as v_pk_*_f32 (the product writes it out per ray)
# -> build/<name>/librt_amd.so, used through RT_AMD_LIB=<path> (raytracer_amd/abi.py).  Never the
# library the tests or the bench load.
set -e
NAME=$1; shift
cd "$(dirname "$0")/../raytracer_amd/csrc"
O=../../build/$NAME
mkdir -p $O
HIPFLAGS="$* --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize -Wno-unused-function -Wno-bitwise-instead-of-logical -Wno-unused-value"
mkdir -p $O/tmp
/opt/rocm/bin/hipcc $HIPFLAGS -save-temps=obj -c rt_kernels.hip -o $O/tmp/rt_kernels.o
python3 ../../tools/isa_spill_lint.py $O/tmp/rt_kernels-hip-amdgcn-amd-amdhsa-gfx950.s || echo "WARNING: this variant has spill code under a partial lane mask (tools/isa_spill_lint.py): its frames cannot be trusted"
cp $O/tmp/rt_kernels.o $O/rt_kernels.o
/opt/rocm/bin/hipcc $HIPFLAGS -c rt_context.hip -o $O/rt_context.o
g++ -O2 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -pthread -c rt_upload.cpp -o $O/rt_upload.o
g++ -O2 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -pthread -c rt_hostio.cpp -o $O/rt_hostio.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/librt_amd.so $O/rt_kernels.o $O/rt_context.o $O/rt_upload.o $O/rt_hostio.o -pthread
echo built $O/librt_amd.so
