#!/bin/bash
# instruction-cache and wait counters of bench.py:  bash tools/r03_icache.sh <tag> [config ...]
set -e
TAG=$1; shift
CONFIGS=${@:-c3}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$TAG
mkdir -p $O
IC=SQC_ICACHE_REQ,SQC_ICACHE_HITS,SQC_ICACHE_MISSES,SQC_ICACHE_MISSES_DUPLICATE,SQC_ICACHE_BUSY_CYCLES,SQC_TC_INST_REQ,SQC_TC_STALL
WT=SQ_WAVE_CYCLES,SQ_BUSY_CYCLES,SQ_WAIT_ANY,SQ_WAIT_INST_ANY,SQ_WAIT_INST_LDS,SQ_IFETCH,SQ_ACTIVE_INST_ANY,SQ_INSTS_VALU
for c in $CONFIGS; do
  case $c in
    c4) A="--config c4 --spp 64 --steps 1 --warmup 1 --no-cpu";;
    c5) A="--config c5 --spp 16 --steps 1 --warmup 1 --no-cpu";;
    *)  A="--config $c --steps 1 --warmup 1 --no-cpu";;
  esac
  timeout -s KILL 300 rocprofv3 --pmc $IC --output-format csv -d $O/ic -o $c -- python3 bench.py $A > $O/ic_$c.log 2>&1
  timeout -s KILL 300 rocprofv3 --pmc $WT --output-format csv -d $O/wt -o $c -- python3 bench.py $A > $O/wt_$c.log 2>&1
  python3 tools/pmc_summary.py k_batch,k_path $O/ic/${c}_counter_collection.csv $O/wt/${c}_counter_collection.csv > $O/icache_$c.txt
done
rm -rf $O/ic $O/wt
