#!/bin/bash
# Instruction and cycle counters of k_batch over the whole frame and over one rank of eight (same process, launches in order):
#   bash tools/r03_pmc_partition.sh <tag> [config] [spp]
TAG=$1; CFG=${2:-c3}; SPP=${3:-64}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$TAG; mkdir -p $O
P1=${P1:-SQ_INSTS_VALU,SQ_INSTS_SALU,SQ_INSTS_SMEM,SQ_WAVE_CYCLES,SQ_BUSY_CYCLES,SQ_WAVES,SQ_ACTIVE_INST_VALU,SQ_WAIT_INST_ANY}
P2=${P2:-SQ_INSTS_VALU,SQ_INSTS_LDS,SQ_INSTS_VMEM_RD,SQ_INSTS_VMEM_WR,SQ_THREAD_CYCLES_VALU,SQ_INST_CYCLES_SMEM,SQ_WAIT_ANY,SQ_ACTIVE_INST_ANY}
n=1
for ctrs in $P1 $P2; do
  REHEARSAL_WORLDS=1,8 timeout -s KILL 300 rocprofv3 --pmc $ctrs --output-format csv -d $O/p$n -o r -- python3 tools/partition_rehearsal.py $CFG $SPP 2 > $O/p$n.log 2>&1 || exit 1
  n=$((n+1))
done
python3 - $O <<'PY'
import csv, sys, collections
O = sys.argv[1]
for p in ("p1", "p2"):
    rows = list(csv.DictReader(open(f"{O}/{p}/r_counter_collection.csv")))
    # launches of k_batch in dispatch order; per dispatch: counter -> value
    disp = collections.OrderedDict()
    for r in rows:
        if "k_batch" not in r["Kernel_Name"]:
            continue
        d = disp.setdefault(int(r["Dispatch_Id"]), {})
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    ids = sorted(disp)
    # partition_rehearsal: world 1 = 1 warm-up + 2 steps; world 8 = two ranks x (1 + 2)
    whole, part = disp[ids[2]], disp[ids[-1]]
    for k in sorted(whole):
        print(f"{k:28s} whole {whole[k]:16.0f}   eighth {part[k]:16.0f}   eighth x 8 / whole {8 * part[k] / max(whole[k], 1):6.3f}")
PY
rm -rf $O/p1 $O/p2
