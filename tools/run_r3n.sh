#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r3n; mkdir -p $O
cd $R
SQ_EQ_CHAIN=1 SQ_CHAIN_BLOCKS=128 timeout -k 5 200 python bench.py --steps 8 --warmup 1 --cpu-sample 0 --fastq-pairs 0 > $O/c2_chain128.json 2> $O/c2_chain128.err
SQ_EQ_CHAIN=1 SQ_CHAIN_BLOCKS=32 timeout -k 5 200 python bench.py --steps 8 --warmup 1 --cpu-sample 0 --fastq-pairs 0 > $O/c2_chain32.json 2> $O/c2_chain32.err
