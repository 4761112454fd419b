#!/bin/bash
# round 4, call F: long reads / uni-MEM overflow / alignment mode on the GPU, the whole suite, hardware-queue experiment for the tail stall
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
cd $R; mkdir -p gpurun_out/r4f; O=$R/gpurun_out/r4f
timeout 900 python -m pytest tests/test_long_reads.py tests/test_alignment_mode.py -m gpu -q > $O/pytest_new.log 2>&1
timeout 1200 python -m pytest tests -m gpu -x -q --deselect tests/test_long_reads.py --deselect tests/test_alignment_mode.py > $O/pytest_gpu.log 2>&1
X="--steps 20 --warmup 1 --no-extras --cpu-sample 0 --fastq-pairs 0 --index-cache /tmp/ixc"
SQ_TIMING=1 timeout 300 python bench.py $X > $O/b_q4.json 2> $O/b_q4.err
SQ_TIMING=1 GPU_MAX_HW_QUEUES=8 timeout 300 python bench.py $X > $O/b_q8.json 2> $O/b_q8.err
SQ_TIMING=1 GPU_MAX_HW_QUEUES=16 timeout 300 python bench.py $X > $O/b_q16.json 2> $O/b_q16.err
SQ_TIMING=1 GPU_MAX_HW_QUEUES=8 timeout 300 python bench.py $X > $O/b_q8b.json 2> $O/b_q8b.err
echo done
