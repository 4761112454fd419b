#!/bin/bash
# round 4, call T: the EM tests on the final library; kernel trace of configs[3] at full size with the flat large-end chaining
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r4t; mkdir -p $O; cd $R
timeout 300 python -m pytest tests/test_em.py -m gpu -x -q > $O/pytest_em.log 2>&1; tail -2 $O/pytest_em.log
cd /tmp
timeout -k 5 480 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python $R/bench.py --workload c4 --genome-gnt 3.1 --steps 3 --warmup 1 --no-extras --cpu-sample 0 --fastq-pairs 0 > $O/kt_c4.json 2> $O/kt_c4.err
db=$(find $O/kt -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/kstats.py $db "" 70 > $O/kernel_stats_c4_full_flat.txt; rm -rf $O/kt
echo done
