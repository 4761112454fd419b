#!/bin/bash
# round 5, call C: after the clean-up, the one-launch selection (k_select with look-back) and the three-launch EM iteration — GPU suite, bench, kernel trace
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r5c; mkdir -p $O; cd $R
python -c "import torch" > /dev/null 2>&1
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_gpu_all.log 2>&1; tail -3 $O/pytest_gpu_all.log
timeout 600 python bench.py --steps 20 --warmup 5 --index-cache /tmp/ixc --no-extras --cpu-sample 0 --fastq-pairs 0 > $O/bench_c2_noextras.json 2> $O/bench_c2_noextras.err; tail -c 300 $O/bench_c2_noextras.json
cd /tmp
Q="--no-extras --cpu-sample 0 --fastq-pairs 0 --index-cache /tmp/ixc"
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python $R/bench.py --steps 10 --warmup 2 $Q > $O/kt_bench.json 2> $O/kt.err
db=$(find $O/kt -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/kstats.py $db "" 60 > $O/kernel_stats_c2.txt; rm -rf $O/kt
head -30 $O/kernel_stats_c2.txt | cut -c1-150
echo done
