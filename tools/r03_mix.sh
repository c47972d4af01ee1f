#!/bin/bash
# Dynamic instruction mix of the render kernels, run on the GPU box:   bash tools/r03_mix.sh <tag> <lib or -> [config ...]
# Per-class SQ instruction counters of bench.py (one pass each, as the guide prescribes), and the same counters on
# tools/valu_microbench classes, which shows what each counter counts (a packed op counts once in its class).
set -e
TAG=$1; LIB=$2; shift; shift
CONFIGS=${@:-c3}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
[ "$LIB" != "-" ] && export RT_AMD_LIB=$LIB
O=gpurun_out/$TAG
mkdir -p $O
MIX1=SQ_INSTS_VALU,SQ_INSTS_VALU_ADD_F32,SQ_INSTS_VALU_MUL_F32,SQ_INSTS_VALU_FMA_F32,SQ_INSTS_VALU_TRANS_F32,SQ_INSTS_VALU_INT32,SQ_INSTS_VALU_CVT,SQ_INSTS_SALU
MIX2=SQ_INSTS_VALU,SQ_ACTIVE_INST_VALU,SQ_INSTS_VMEM_RD,SQ_INSTS_VMEM_WR,SQ_BUSY_CYCLES,SQ_WAVE_CYCLES,SQ_INSTS_SMEM,SQ_INSTS_LDS
MIX3=SQ_INSTS_VALU,SQ_INSTS_VALU_ADD_F64,SQ_INSTS_VALU_MUL_F64,SQ_INSTS_VALU_FMA_F64,SQ_INSTS_VALU_INT64,SQ_INSTS_BRANCH,SQ_INSTS_VALU_IOPS,SQ_INSTS_VSKIPPED
for c in $CONFIGS; do
  case $c in
    c4) A="--config c4 --spp 64 --steps 1 --warmup 1 --no-cpu";;
    c5) A="--config c5 --spp 64 --steps 1 --warmup 1 --no-cpu";;
    *)  A="--config $c --steps 1 --warmup 1 --no-cpu";;
  esac
  n=1
  for ctrs in $MIX1 $MIX2 $MIX3; do
    echo "== $c pmc mix$n"; date +%T
    timeout -s KILL 300 rocprofv3 --pmc $ctrs --output-format csv -d $O/mix$n -o $c -- python3 bench.py $A > $O/mix${n}_$c.log 2>&1
    n=$((n+1))
  done
  python3 tools/pmc_summary.py k_batch,k_path $O/mix1/${c}_counter_collection.csv $O/mix2/${c}_counter_collection.csv $O/mix3/${c}_counter_collection.csv > $O/mix_$c.txt
done
rm -rf $O/mix1 $O/mix2 $O/mix3
if [ -n "$MB" ]; then
  for k in v_and_b32 v_mov_b32 v_cndmask v_max_f32 "v_cmp_gt_f32 -> vcc" v_add_u32 v_lshrrev v_or3 v_readlane v_mul_lo v_bfe v_cvt; do
    t=$(echo "$k" | tr ' />-' '____')
    timeout -s KILL 60 rocprofv3 --pmc $MIX1 --output-format csv -d $O/mb_$t -o mb -- build/valu_microbench /dev/null "$k" > $O/mb_$t.log 2>&1
    timeout -s KILL 60 rocprofv3 --pmc $MIX3 --output-format csv -d $O/mc_$t -o mb -- build/valu_microbench /dev/null "$k" > $O/mc_$t.log 2>&1
    python3 tools/pmc_summary.py k_bench $O/mb_$t/mb_counter_collection.csv $O/mc_$t/mb_counter_collection.csv > $O/mb_$t.txt || true
    rm -rf $O/mb_$t $O/mc_$t
  done
fi
echo done
