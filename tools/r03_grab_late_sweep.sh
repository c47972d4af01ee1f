#!/bin/bash
# units per ticket of the late units' dispensers (RT_GRAB_LATE): whole frame and an eighth of C3, C4 an eighth
for rep in 1 2; do
for g in 4 2 8 16; do
  echo "== RT_GRAB_LATE=$g (run $rep)"
  RT_GRAB_LATE=$g REHEARSAL_WORLDS=1,8 timeout -k 10 300 python tools/partition_rehearsal.py c3 64 10 || exit 1
done
done
for g in 4 8 16; do
  echo "== RT_GRAB_LATE=$g c4"
  RT_GRAB_LATE=$g REHEARSAL_WORLDS=1,8 timeout -k 10 300 python tools/partition_rehearsal.py c4 64 4 || exit 1
done
