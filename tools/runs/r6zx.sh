#!/bin/bash
# round 6, call ZX: SQ / TCP counters of the eq stream's kernels after burn-in (six timed steps: the split static / dynamic path runs from the third)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r6zx; mkdir -p $O; cd $R
python -c "import torch" > /dev/null 2>&1
cd /tmp
for c in "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum"; do
  tag=$(echo $c | cut -d' ' -f1)
  timeout -k 5 400 rocprofv3 --kernel-trace --pmc $c -d $O/pmc_$tag -o pmc --output-format csv -- python $R/bench.py --steps 6 --warmup 0 --no-extras --cpu-sample 0 --fastq-pairs 0 --index-cache /tmp/ixc > /dev/null 2> $O/pmc_$tag.err
  python $R/tools/pmc_summary.py $O/pmc_$tag 80 $O/pmc_$tag.json > $O/pmc_$tag.txt; rm -rf $O/pmc_$tag
  grep -E "^k_frag_static|^k_frag_dynamic|^k_apply_flagged|^k_eq_insert|^k_eq_add|^k_pre_aln" $O/pmc_$tag.txt | cut -c1-300
done
echo done
