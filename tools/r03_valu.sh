#!/bin/bash
# Round-3 VALU accounting, run on the GPU box from the repo root:   bash tools/r03_valu.sh <tag> [config ...]
#  1. tools/valu_microbench: SIMD cycles per instruction of each class at 1..8 waves per SIMD
#  2. bench.py per config under the per-class instruction counters: the dynamic instruction mix of the kernels
set -e
TAG=${1:-valu}
shift || true
CONFIGS=${@:-c3}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$TAG
mkdir -p $O
MIX1=SQ_INSTS_VALU,SQ_INSTS_VALU_ADD_F32,SQ_INSTS_VALU_MUL_F32,SQ_INSTS_VALU_FMA_F32,SQ_INSTS_VALU_TRANS_F32,SQ_INSTS_VALU_INT32,SQ_INSTS_VALU_CVT,SQ_INSTS_SALU
MIX2=SQ_INSTS_VALU,SQ_ACTIVE_INST_VALU,SQ_INSTS_VMEM_RD,SQ_INSTS_VMEM_WR,SQ_BUSY_CYCLES,SQ_WAVE_CYCLES,SQ_INSTS_SMEM,SQ_INSTS_LDS
echo "== microbench"; date +%T
timeout -k 10 200 build/valu_microbench $O/microbench.csv > $O/microbench.txt 2>&1
for c in $CONFIGS; do
  case $c in
    c4) A="--config c4 --spp 64 --steps 1 --warmup 1 --no-cpu";;
    c5) A="--config c5 --spp 64 --steps 1 --warmup 1 --no-cpu";;
    *)  A="--config $c --steps 1 --warmup 1 --no-cpu";;
  esac
  echo "== $c pmc mix"; date +%T
  timeout -s KILL 300 rocprofv3 --pmc $MIX1 --output-format csv -d $O/mix1 -o $c -- python3 bench.py $A > $O/mix1_$c.log 2>&1
  timeout -s KILL 300 rocprofv3 --pmc $MIX2 --output-format csv -d $O/mix2 -o $c -- python3 bench.py $A > $O/mix2_$c.log 2>&1
  python3 tools/pmc_summary.py k_batch,k_path $O/mix1/${c}_counter_collection.csv $O/mix2/${c}_counter_collection.csv > $O/mix_$c.txt
done
rm -rf $O/mix1 $O/mix2
echo done
