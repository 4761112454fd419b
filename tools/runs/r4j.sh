#!/bin/bash
# round 4, call J: per-group transcript flags for the mass application (byte stores + LDS compaction), eq partition size
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
cd $R; mkdir -p gpurun_out/r4j; O=$R/gpurun_out/r4j
X="--steps 20 --warmup 1 --no-extras --cpu-sample 0 --fastq-pairs 0 --index-cache /tmp/ixc"
timeout 300 python bench.py $X > $O/b_flags64.json 2> $O/b_flags64.err
SQ_EQ_CUS=48 timeout 300 python bench.py $X > $O/b_flags48.json 2> $O/b_flags48.err
SQ_EQ_CUS=56 timeout 300 python bench.py $X > $O/b_flags56.json 2> $O/b_flags56.err
SQ_EQ_CUS=40 timeout 300 python bench.py $X > $O/b_flags40.json 2> $O/b_flags40.err
SQ_EQ_SPLIT=1 SQ_EQ_CUS=32 timeout 300 python bench.py $X > $O/b_split32.json 2> $O/b_split32.err
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_all.log 2>&1
echo done
