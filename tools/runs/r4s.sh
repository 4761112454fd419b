#!/bin/bash
# round 4, call S: the EM's first upload by a kernel (SQ_EM_KUP=1, new default) against a copy command (0): the drain that follows it, four runs each
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r4s; mkdir -p $O; cd $R
X="--steps 20 --warmup 1 --no-extras --cpu-sample 0 --fastq-pairs 0 --index-cache /tmp/ixc"
timeout 300 python bench.py --steps 2 --warmup 1 --no-extras --cpu-sample 0 --fastq-pairs 0 --index-cache /tmp/ixc > /dev/null 2>&1
for i in 1 2 3 4; do
SQ_TIMING=1 timeout 300 python bench.py $X > $O/b_kup_$i.json 2> $O/b_kup_$i.err
SQ_TIMING=1 SQ_EM_KUP=0 timeout 300 python bench.py $X > $O/b_cmd_$i.json 2> $O/b_cmd_$i.err
done
echo done
