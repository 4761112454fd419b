#!/bin/bash
# round 4, call L: device reader with two batches in flight; kernel trace of configs[3] at full size (what the large-end stage spends its 30 ms on)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
cd $R; mkdir -p gpurun_out/r4l; O=$R/gpurun_out/r4l
timeout 300 python -m pytest tests/test_reader_gpu.py -m gpu -x -q > $O/pytest_reader.log 2>&1
SQ_READER_STATS=1 timeout 400 python bench.py --steps 8 --warmup 1 --no-extras --cpu-sample 0 --fastq-pairs 40000000 --index-cache /tmp/ixc > $O/b_fastq.json 2> $O/b_fastq.err
cd /tmp
timeout -k 5 1500 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python $R/bench.py --workload c4 --genome-gnt 3.1 --steps 3 --warmup 1 --no-extras --cpu-sample 0 --fastq-pairs 0 > $O/kt_c4.json 2> $O/kt_c4.err
db=$(find $O/kt -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/kstats.py $db "" 70 > $O/kernel_stats_c4_full.txt; rm -rf $O/kt
echo done
