#!/bin/bash
# Round profile, run on the GPU box from the repo root:  bash tools/profile_round.sh <tag>
# Writes gpurun_out/<tag>/: bench lines (c2, c3), rocprofv3 kernel-trace stats, and the PMC
# passes (each its own run, as MI355X_MICROARCH.md prescribes): FETCH_SIZE, WRITE_SIZE, and
# two SQ passes (issue / wait breakdown; scalar-cache behaviour).  tools/pmc_summary.py turns
# them into the summaries committed under profiles/.
set -e
TAG=${1:-prof}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$TAG
mkdir -p $O
timeout -k 10 300 python bench.py --config c3 --steps 3 --warmup 1 > $O/bench_c3.json 2> $O/bench_c3.err
timeout -k 10 300 python bench.py --config c2 --steps 5 --warmup 2 > $O/bench_c2.json 2> $O/bench_c2.err
SQ1=SQ_WAVE_CYCLES,SQ_BUSY_CYCLES,SQ_WAIT_ANY,SQ_WAIT_INST_ANY,SQ_ACTIVE_INST_ANY,SQ_ACTIVE_INST_VALU,SQ_ACTIVE_INST_SCA,SQ_INSTS_VALU
SQ2=SQ_INSTS_SALU,SQ_INSTS_SMEM,SQ_INST_CYCLES_SMEM,SQ_INSTS_LDS,SQ_WAVES,SQ_INSTS_VMEM_RD,SQ_INSTS_VMEM_WR,SQ_CYCLES
SQC=SQC_DCACHE_REQ,SQC_DCACHE_HITS,SQC_DCACHE_MISSES,SQC_DCACHE_MISSES_DUPLICATE,SQC_TC_DATA_READ_REQ,SQC_TC_STALL,SQC_DCACHE_BUSY_CYCLES
for c in c3 c2; do
  if [ $c = c2 ]; then A="--config c2 --steps 2 --warmup 1 --no-cpu"; else A="--config c3 --steps 1 --warmup 1 --no-cpu"; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o $c -- python3 bench.py $A > $O/kt_$c.log 2>&1
  timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o $c -- python3 bench.py $A > $O/fetch_$c.log 2>&1
  timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o $c -- python3 bench.py $A > $O/write_$c.log 2>&1
  timeout -s KILL 120 rocprofv3 --pmc $SQ1 --output-format csv -d $O/sq1 -o $c -- python3 bench.py $A > $O/sq1_$c.log 2>&1
  timeout -s KILL 120 rocprofv3 --pmc $SQ2 --output-format csv -d $O/sq2 -o $c -- python3 bench.py $A > $O/sq2_$c.log 2>&1
  timeout -s KILL 120 rocprofv3 --pmc $SQC --output-format csv -d $O/sqc -o $c -- python3 bench.py $A > $O/sqc_$c.log 2>&1
  python3 tools/pmc_traffic.py $c $O/fetch/${c}_counter_collection.csv $O/write/${c}_counter_collection.csv $O/traffic.json > $O/traffic_$c.txt
  python3 tools/pmc_summary.py k_shadow,k_primary,k_primary_hard,k_path $O/sq1/${c}_counter_collection.csv $O/sq2/${c}_counter_collection.csv $O/sqc/${c}_counter_collection.csv > $O/sq_$c.txt
done
echo done
