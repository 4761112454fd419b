#!/bin/bash
# round 6, call ZN: the two --mimicBT2 / --mimicStrictBT2 option variants against the checker; what k_pack waits for (SQ counters over the c2 bench, two batches)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r6zn; mkdir -p $O; cd $R
python -c "import torch" > /dev/null 2>&1
timeout -k 5 600 python -m pytest tests/test_map_gpu.py -m gpu -x -q -k "mimic or hard_filter" > $O/gputests_mimic.txt 2>&1; grep -E "passed|failed|error" $O/gputests_mimic.txt | tail -3
cd /tmp
for c in "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum"; do
  tag=$(echo $c | cut -d' ' -f1)
  timeout -k 5 300 rocprofv3 --kernel-trace --pmc $c -d $O/pmc_$tag -o pmc --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-extras --cpu-sample 0 --fastq-pairs 0 --index-cache /tmp/ixc > /dev/null 2> $O/pmc_$tag.err
  python $R/tools/pmc_summary.py $O/pmc_$tag 60 $O/pmc_$tag.json > $O/pmc_$tag.txt; rm -rf $O/pmc_$tag
  grep -E "k_pack|k_select|k_join2 |k_score|k_seed2" $O/pmc_$tag.txt | head -40
done
echo done
