#!/bin/bash
# round 5, call L (final): the whole GPU suite, the driver's bench command, kernel trace, FETCH_SIZE / WRITE_SIZE passes
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r5l; mkdir -p $O; cd $R
python -c "import torch" > /dev/null 2>&1
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_gpu_all.log 2>&1; tail -3 $O/pytest_gpu_all.log
timeout 600 python bench.py --steps 20 --warmup 5 --index-cache /tmp/ixc > $O/bench_c2_default.json 2> $O/bench_c2_default.err; tail -c 600 $O/bench_c2_default.json
cd /tmp
Q="--no-extras --cpu-sample 0 --fastq-pairs 0 --index-cache /tmp/ixc"
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python $R/bench.py --steps 10 --warmup 2 $Q > $O/kt_bench.json 2> $O/kt.err
db=$(find $O/kt -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/kstats.py $db "" 60 > $O/kernel_stats_c2.txt; rm -rf $O/kt
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 5 240 rocprofv3 --kernel-trace --pmc $c -d $O/pmc_$c -o pmc --output-format csv -- python $R/bench.py --steps 2 --warmup 1 $Q > /dev/null 2> $O/pmc_$c.err
  python $R/tools/pmc_summary.py $O/pmc_$c 60 $O/pmc_$c.json > $O/pmc_$c.txt; rm -rf $O/pmc_$c
done
head -12 $O/kernel_stats_c2.txt
echo done
