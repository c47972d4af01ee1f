#!/bin/bash
# build/stamps/librt_amd.so: the library with in-kernel s_memtime stamps (tools/stamps_report.py)
exec bash "$(dirname "$0")/build_variant.sh" stamps -DRT_STAMPS $EXTRA
