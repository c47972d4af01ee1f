#!/bin/bash
# knobs of k_batch's dispatch on an eighth of C3 (final build):  bash tools/r03_eighth_sweep.sh
one() { echo "== $*"; env "$@" REHEARSAL_WORLDS=8 timeout -k 10 200 python tools/partition_rehearsal.py c3 64 12 | grep "world 8" || exit 1; }
for rep in 1 2; do
one RT_NONE=0
one RT_UNITS_PER_BLOCK=8
one RT_UNITS_PER_BLOCK=32
one RT_EARLY_X4=6
one RT_EARLY_X4=12
one RT_LATE_X16=4
one RT_LATE_X16=16
one RT_GRAB_LATE=2
one RT_GRAB_LATE=8
one RT_TASK_PRIO=0
done
