#!/bin/bash
# round 4, call I: touched-transcript lists for the chain (old 192/64 layout and the split layout), the device FASTQ reader
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
cd $R; mkdir -p gpurun_out/r4i; O=$R/gpurun_out/r4i
timeout 600 python -m pytest tests/test_reader_gpu.py -m gpu -x -q > $O/pytest_first.log 2>&1
X="--steps 20 --warmup 1 --no-extras --cpu-sample 0 --fastq-pairs 0 --index-cache /tmp/ixc"
timeout 300 python bench.py $X > $O/b_old_touched.json 2> $O/b_old_touched.err
SQ_EQ_TOUCHED=0 timeout 300 python bench.py $X > $O/b_old_sweep.json 2> $O/b_old_sweep.err
SQ_EQ_SPLIT=1 timeout 300 python bench.py $X > $O/b_split16.json 2> $O/b_split16.err
SQ_EQ_SPLIT=1 SQ_EQ_CUS=24 timeout 300 python bench.py $X > $O/b_split24.json 2> $O/b_split24.err
SQ_EQ_SPLIT=1 SQ_EQ_CUS=32 timeout 300 python bench.py $X > $O/b_split32.json 2> $O/b_split32.err
timeout 400 python bench.py --steps 8 --warmup 1 --no-extras --cpu-sample 0 --fastq-pairs 40000000 --index-cache /tmp/ixc > $O/b_fastq.json 2> $O/b_fastq.err
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_all.log 2>&1
echo done
