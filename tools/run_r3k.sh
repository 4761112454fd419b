#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r3k; mkdir -p $O
cd $R
SQ_EQ_PRIO=1 timeout 400 python bench.py --steps 8 --warmup 1 --cpu-sample 200000 --fastq-pairs 0 > $O/c2_prio.json 2> $O/c2_prio.err
SQ_EQ_PRIO=1 SQ_SEED_BPC=5 timeout 400 python bench.py --steps 8 --warmup 1 --cpu-sample 0 --fastq-pairs 0 > $O/c2_prio_b5.json 2> $O/c2_prio_b5.err
