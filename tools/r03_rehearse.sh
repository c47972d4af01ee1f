#!/bin/bash
# bash tools/r03_rehearse.sh <out-dir> <config> <spp> <worlds> VAR=val[,VAR=val] ...   (one rehearsal per knob set, interleaved twice)
O=$1; CFG=$2; SPP=$3; W=$4; shift 4
mkdir -p $O
for rep in 1 2; do
  for knobs in "$@"; do
    tag=$(echo "$knobs" | tr ',=/' '___')
    echo "== $knobs (run $rep)"
    env $(echo "$knobs" | tr ',' ' ') REHEARSAL_WORLDS=$W timeout -k 10 300 python tools/partition_rehearsal.py $CFG $SPP 5 2>&1 | tee -a $O/${CFG}_${tag}.txt
  done
done
