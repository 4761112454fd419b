#!/bin/bash
# round 5, call K: where the device inflater's time goes — SQ counters of k_bgzf_inflate on random-quality FASTQ members (level 1), a few counters per pass
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r5k; mkdir -p $O; cd /tmp
python -c "import torch" > /dev/null 2>&1
i=0
for grp in "SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_FLAT" "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS"; do
  i=$((i+1))
  timeout -k 5 200 rocprofv3 --kernel-trace --pmc $grp -d $O/p$i -o pmc --output-format csv -- python $R/tools/inflate_bench.py 0,1 > $O/p$i.log 2>&1
  f=$(find $O/p$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python - "$f" <<'PY' | tee -a $O/counters.txt
import csv, sys, collections
acc = collections.defaultdict(float); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if "inflate" not in r.get("Kernel_Name", ""): continue
    acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in acc: print("%-28s %16.0f per launch (%d rows)" % (k, acc[k] / 3.0, n[k]))
PY
  else echo "group $i: no counters ($(tail -2 $O/p$i.log | tr '\n' ' '))" | tee -a $O/counters.txt; fi
  rm -rf $O/p$i
done
echo done
