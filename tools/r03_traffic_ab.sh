#!/bin/bash
# HBM traffic of k_batch per launch (FETCH_SIZE / WRITE_SIZE passes) for a few settings:  bash tools/r03_traffic_ab.sh <tag> <config> <steps>
TAG=$1; C=${2:-c3}; STEPS=${3:-18}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$TAG; mkdir -p $O
run() {  # name, env...
  name=$1; shift
  for p in fetch:FETCH_SIZE write:WRITE_SIZE; do
    n=${p%%:*}; ctr=${p#*:}
    env "$@" true
    ( export "$@"; timeout -s KILL 300 rocprofv3 --pmc $ctr --output-format csv -d $O/$name-$n -o x -- python3 bench.py --config $C --steps $STEPS --warmup 2 --no-cpu > $O/$name-$n.log 2>&1 ) || return 1
  done
  echo "== $name"; python3 tools/pmc_traffic.py $C $O/$name-fetch/x_counter_collection.csv $O/$name-write/x_counter_collection.csv $O/$name.json | grep k_batch
  rm -rf $O/$name-fetch $O/$name-write
}
run default RT_DUMMY=1 || exit 1
run grab1 RT_GRAB=1 || exit 1
run nofeedback RT_GRAB=1 RT_FEEDBACK=0 || exit 1
run prevlib RT_AMD_LIB=build/old/librt_amd.so || exit 1
