#!/bin/bash
# round 4, call H: the eq stage split (throughput kernels on the mapping pool, the chain on a small partition) against the 192/64 layout
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
cd $R; mkdir -p gpurun_out/r4h; O=$R/gpurun_out/r4h
X="--steps 20 --warmup 1 --no-extras --cpu-sample 0 --fastq-pairs 0 --index-cache /tmp/ixc"
timeout 300 python bench.py $X > $O/b_old.json 2> $O/b_old.err
SQ_EQ_SPLIT=1 timeout 300 python bench.py $X > $O/b_split16.json 2> $O/b_split16.err
SQ_EQ_SPLIT=1 SQ_EQ_CUS=24 timeout 300 python bench.py $X > $O/b_split24.json 2> $O/b_split24.err
SQ_EQ_SPLIT=1 SQ_EQ_CUS=32 timeout 300 python bench.py $X > $O/b_split32.json 2> $O/b_split32.err
SQ_EQ_SPLIT=1 SQ_EQ_CUS=8 timeout 300 python bench.py $X > $O/b_split8.json 2> $O/b_split8.err
SQ_EQ_SPLIT=1 timeout 300 python bench.py $X --workload c2s > $O/b_c2s_split16.json 2> $O/b_c2s_split16.err
timeout 300 python bench.py $X --workload c2s > $O/b_c2s_old.json 2> $O/b_c2s_old.err
SQ_EQ_SPLIT=1 timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_split.log 2>&1
echo done
