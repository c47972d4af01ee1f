#!/bin/bash
# Where k_bounce's time goes (bounce queue on, C3 at depth 12):   bash tools/r03_pmc_bounce.sh <tag>
TAG=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$TAG; mkdir -p $O
export RT_BOUNCE_QUEUE=${RT_BOUNCE_QUEUE:-1} BOUNCE_CONFIGS=c3
P1=TA_BUSY_avr,TA_FLAT_READ_WAVEFRONTS_sum,GRBM_GUI_ACTIVE,SQ_INSTS_VMEM_RD,SQ_WAVE_CYCLES,SQ_BUSY_CYCLES,SQ_INSTS_VALU,SQ_ACTIVE_INST_VALU,SQ_WAIT_ANY,SQ_WAIT_INST_ANY,SQ_WAVES
P2=TA_ADDR_STALLED_BY_TC_CYCLES_sum,TA_DATA_STALLED_BY_TC_CYCLES_sum,TCP_TOTAL_CACHE_ACCESSES_sum,TCP_TCC_READ_REQ_sum,TCP_PENDING_STALL_CYCLES_sum,TCP_TCP_TA_DATA_STALL_CYCLES_sum,SQ_THREAD_CYCLES_VALU,SQ_ACTIVE_INST_VALU,SQ_INSTS_SALU,SQ_INST_CYCLES_VMEM_RD
n=1
for ctrs in $P1 $P2; do
  timeout -s KILL 200 rocprofv3 --pmc $ctrs --output-format csv -d $O/p$n -o r -- python3 tools/bounce_bench.py 12 4 2 > $O/p$n.log 2>&1 || { tail -5 $O/p$n.log; exit 1; }
  n=$((n+1))
done
python3 tools/pmc_summary.py k_bounce,k_bounce_sm,k_path $O/p1/r_counter_collection.csv $O/p2/r_counter_collection.csv > $O/summary.txt
rm -rf $O/p1 $O/p2
cat $O/summary.txt
