#!/bin/bash
# Round profile, run on the GPU box from the repo root:
#     bash tools/profile_round.sh <tag> [config ...]        (default configs: c3 c2)
# Writes gpurun_out/<tag>/: one bench line per config, rocprofv3 kernel-trace stats, and the PMC
# passes (each its own run, as MI355X_MICROARCH.md prescribes): FETCH_SIZE, WRITE_SIZE, L2 hit /
# miss, three SQ passes (issue / wait breakdown; instruction mix; lane utilisation =
# SQ_THREAD_CYCLES_VALU / (64 SQ_ACTIVE_INST_VALU)) and the scalar cache.  tools/pmc_summary.py
# and tools/pmc_traffic.py turn them into the summaries committed under profiles/.
set -e
TAG=${1:-prof}
shift || true
CONFIGS=${@:-c3 c2}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$TAG
mkdir -p $O
SQ1=SQ_WAVE_CYCLES,SQ_BUSY_CYCLES,SQ_WAIT_ANY,SQ_WAIT_INST_ANY,SQ_ACTIVE_INST_ANY,SQ_ACTIVE_INST_VALU,SQ_ACTIVE_INST_SCA,SQ_INSTS_VALU
SQ2=SQ_INSTS_SALU,SQ_INSTS_SMEM,SQ_INST_CYCLES_SMEM,SQ_INSTS_LDS,SQ_WAVES,SQ_INSTS_VMEM_RD,SQ_INSTS_VMEM_WR,SQ_CYCLES
SQ3=SQ_THREAD_CYCLES_VALU,SQ_ACTIVE_INST_VALU,SQ_INST_CYCLES_VALU,SQ_INSTS_VALU
SQC=SQC_DCACHE_REQ,SQC_DCACHE_HITS,SQC_DCACHE_MISSES,SQC_DCACHE_MISSES_DUPLICATE,SQC_TC_DATA_READ_REQ,SQC_TC_STALL,SQC_DCACHE_BUSY_CYCLES
L2=TCC_HIT_sum,TCC_MISS_sum,TCC_REQ_sum
for c in $CONFIGS; do
  case $c in
    # (k_batch's feedback settles with the third launch of a shape: two warm-up steps everywhere, and enough timed
    # ones in the traced runs that the two slower launches weigh little in the per-kernel averages)
    c2) B="--config c2 --steps 5 --warmup 2";           A="--config c2 --steps 5 --warmup 2 --no-cpu";;
    c3) B="--config c3 --steps 5 --warmup 2";           A="--config c3 --steps 18 --warmup 2 --no-cpu";;
    c4) B="--config c4 --steps 3 --warmup 2";           A="--config c4 --steps 6 --warmup 2 --no-cpu";;
    c5) B="--config c5 --steps 2 --warmup 2";           A="--config c5 --steps 2 --warmup 2 --no-cpu";;
    *)  B="--config $c --steps 3 --warmup 1";           A="--config $c --steps 1 --warmup 1 --no-cpu";;
  esac
  echo "== $c: bench"; date +%T
  timeout -k 10 400 python bench.py $B > $O/bench_$c.json 2> $O/bench_$c.err
  echo "== $c: kernel trace"; date +%T
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o $c -- python3 bench.py $A > $O/kt_$c.log 2>&1
  for pass in fetch:FETCH_SIZE write:WRITE_SIZE l2:$L2 sq1:$SQ1 sq2:$SQ2 sq3:$SQ3 sqc:$SQC; do
    name=${pass%%:*}; ctrs=${pass#*:}
    echo "== $c: pmc $name"; date +%T
    timeout -s KILL 300 rocprofv3 --pmc $ctrs --output-format csv -d $O/$name -o $c -- python3 bench.py $A > $O/${name}_$c.log 2>&1
  done
  python3 tools/pmc_traffic.py $c $O/fetch/${c}_counter_collection.csv $O/write/${c}_counter_collection.csv $O/traffic.json > $O/traffic_$c.txt
  python3 tools/pmc_summary.py k_batch,k_path,k_intersect,k_fill_i32,k_scatter_image $O/sq1/${c}_counter_collection.csv $O/sq2/${c}_counter_collection.csv $O/sq3/${c}_counter_collection.csv $O/sqc/${c}_counter_collection.csv $O/l2/${c}_counter_collection.csv > $O/sq_$c.txt
  cp $O/kt/${c}_kernel_stats.csv $O/kernel_stats_$c.csv 2>/dev/null || true
  # the raw per-dispatch CSVs are large: keep the summaries only
  rm -rf $O/fetch/${c}_* $O/write/${c}_* $O/l2/${c}_* $O/sq1/${c}_* $O/sq2/${c}_* $O/sq3/${c}_* $O/sqc/${c}_* $O/kt/${c}_*
done
echo done
