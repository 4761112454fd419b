#!/bin/bash
# round 6, call ZH: what the clusters of the flat large-end passes look like on configs[3] at full size (sizes, inner-loop trips of k_lg_dp per cluster); instrumentation is not in the tree
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r6zh; mkdir -p $O; cd $R
python -c "import torch" > /dev/null 2>&1
C4="--workload c4 --genome-gnt 3.1 --warmup 0 --no-extras --fastq-pairs 0 --index-cache /tmp/ixc4"
SQ_LG_STATS=1 timeout 1500 python bench.py $C4 --steps 1 --cpu-sample 0 > $O/bench_c4_stats.json 2> $O/bench_c4_stats.err
grep "lg-stats" $O/bench_c4_stats.err | head -80
echo done
