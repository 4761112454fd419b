#!/bin/bash
# round 4, call C: full GPU test suite, EM iteration with 3 / 4 / 5 launches, phase timings of the tail, kernel trace of the EM loop
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
cd $R; mkdir -p gpurun_out/r4c; O=$R/gpurun_out/r4c
free -g > $O/host.txt; nproc >> $O/host.txt
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1
X="--steps 20 --warmup 1 --no-extras --cpu-sample 0 --fastq-pairs 0 --index-cache /tmp/ixc"
SQ_TIMING=1 timeout 300 python bench.py $X > $O/b_em4.json 2> $O/b_em4.err
SQ_EM_LAUNCHES=5 timeout 300 python bench.py $X > $O/b_em5.json 2> $O/b_em5.err
SQ_EM_LAUNCHES=3 timeout 300 python bench.py $X > $O/b_em3.json 2> $O/b_em3.err
cd /tmp
timeout -k 5 600 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python $R/bench.py --steps 4 --warmup 1 --no-extras --cpu-sample 0 --fastq-pairs 0 --index-cache /tmp/ixc > $O/kt.json 2> $O/kt.err
db=$(find $O/kt -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/kstats.py $db "" 60 > $O/kernel_stats_c2.txt; rm -rf $O/kt
echo done
