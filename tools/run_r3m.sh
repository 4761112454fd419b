#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r3m; mkdir -p $O
cd $R
SQ_EQ_XCD=1 timeout -k 5 300 python bench.py --steps 8 --warmup 1 --cpu-sample 200000 --fastq-pairs 0 > $O/c2_xcd2.json 2> $O/c2_xcd2.err
SQ_EQ_XCD=1 SQ_EQ_CUS=32 timeout -k 5 300 python bench.py --steps 8 --warmup 1 --cpu-sample 0 --fastq-pairs 0 > $O/c2_xcd1.json 2> $O/c2_xcd1.err
