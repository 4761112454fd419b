#!/bin/bash
# round 6, call ZT: the packed read ends against the packing rule (SQ_TAP_PACKED), then the PMC passes again on the tree's final sources
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r6zt; mkdir -p $O; cd $R
python -c "import torch" > /dev/null 2>&1
timeout -k 5 600 python -m pytest tests/test_map_gpu.py -m gpu -x -q -k "packing_rule or ragged or long_reads" > $O/gputests_pack.txt 2>&1; grep -E "passed|failed|error|Error|assert" $O/gputests_pack.txt | tail -8
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 5 300 rocprofv3 --kernel-trace --pmc $c -d $O/pmc_$c -o pmc --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-extras --cpu-sample 0 --fastq-pairs 0 --index-cache /tmp/ixc > /dev/null 2> $O/pmc_$c.err
  python $R/tools/pmc_summary.py $O/pmc_$c 60 $O/pmc_$c.json > $O/pmc_$c.txt; rm -rf $O/pmc_$c
done
head -6 $O/pmc_FETCH_SIZE.txt
echo done
