#!/bin/bash
# round 4, call K: flags vs sweep on one box (the first run of a box is slower: discard it), the two-stage device reader
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
cd $R; mkdir -p gpurun_out/r4k; O=$R/gpurun_out/r4k
X="--steps 20 --warmup 1 --no-extras --cpu-sample 0 --fastq-pairs 0 --index-cache /tmp/ixc"
SQ_EQ_TOUCHED=0 timeout 300 python bench.py $X > $O/b_0_discard.json 2> $O/b_0_discard.err
for i in 1 2; do
timeout 300 python bench.py $X > $O/b_flags_$i.json 2> $O/b_flags_$i.err
SQ_EQ_TOUCHED=0 timeout 300 python bench.py $X > $O/b_sweep_$i.json 2> $O/b_sweep_$i.err
done
timeout 300 python -m pytest tests/test_reader_gpu.py -m gpu -x -q > $O/pytest_reader.log 2>&1
SQ_READER_STATS=1 timeout 400 python bench.py --steps 8 --warmup 1 --no-extras --cpu-sample 0 --fastq-pairs 40000000 --index-cache /tmp/ixc > $O/b_fastq.json 2> $O/b_fastq.err
echo done
