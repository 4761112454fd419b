#!/bin/bash
# round 4, call E: long reads / uni-MEM overflow tests + whole GPU suite, keep-warm on the optimiser's stream
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
cd $R; mkdir -p gpurun_out/r4e; O=$R/gpurun_out/r4e
timeout 900 python -m pytest tests/test_long_reads.py -m gpu -x -q > $O/pytest_long.log 2>&1
timeout 1200 python -m pytest tests -m gpu -x -q --deselect tests/test_long_reads.py > $O/pytest_gpu.log 2>&1
X="--steps 20 --warmup 1 --no-extras --cpu-sample 0 --fastq-pairs 0 --index-cache /tmp/ixc"
SQ_TIMING=1 timeout 300 python bench.py $X > $O/b_warm1.json 2> $O/b_warm1.err
SQ_TIMING=1 SQ_KEEP_WARM=0 timeout 300 python bench.py $X > $O/b_warm0.json 2> $O/b_warm0.err
SQ_TIMING=1 timeout 300 python bench.py $X > $O/b_warm1b.json 2> $O/b_warm1b.err
SQ_TIMING=1 SQ_KEEP_WARM=0 timeout 300 python bench.py $X > $O/b_warm0b.json 2> $O/b_warm0b.err
echo done
