#!/bin/bash
# Diagnostic build of the library with in-kernel s_memtime stamps (-DRT_STAMPS, rt_kernels.hip):
# build/stamps/librt_amd.so, used through RT_AMD_LIB=<path> (raytracer_amd/abi.py) by
# tools/stamps_report.py.  Never the library the tests or the bench load.
set -e
cd "$(dirname "$0")/../raytracer_amd/csrc"
O=../../build/stamps
mkdir -p $O
HIPFLAGS="-DRT_STAMPS $EXTRA --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -Wno-unused-function -Wno-bitwise-instead-of-logical"
/opt/rocm/bin/hipcc $HIPFLAGS -c rt_kernels.hip -o $O/rt_kernels.o
/opt/rocm/bin/hipcc $HIPFLAGS -c rt_context.hip -o $O/rt_context.o
g++ -O2 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -pthread -c rt_upload.cpp -o $O/rt_upload.o
g++ -O2 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -pthread -c rt_hostio.cpp -o $O/rt_hostio.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/librt_amd.so $O/rt_kernels.o $O/rt_context.o $O/rt_upload.o $O/rt_hostio.o -pthread
echo built $O/librt_amd.so
