#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r3end; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_bias_gpu.py tests/test_sampling.py tests/test_dist_gpu.py tests/test_golden.py -m gpu -q --timeout 300 > $O/pytest_c.log 2>&1; echo "pytest rc=$?" >> $O/pytest_c.log
timeout 600 python bench.py --workload c5 > $O/bench_c5.json 2> $O/bench_c5.err
cd /tmp
timeout -k 5 600 rocprofv3 --kernel-trace --stats -d $O/kt_c5 -o kt -- python $R/bench.py --workload c5 --steps 4 --warmup 1 --cpu-sample 0 --fastq-pairs 0 > $O/kt_c5.json 2> $O/kt_c5.err
db=$(find $O/kt_c5 -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/kstats.py $db "" 45 > $O/kernel_stats_c5.txt; rm -rf $O/kt_c5
