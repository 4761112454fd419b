#!/bin/bash
# round 4, call R: the EM's waits by polling (SQ_EM_SPIN=1, the new default) against hipStreamSynchronize (0): the tail of the 100 M-pair job, three runs each
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r4r; mkdir -p $O; cd $R
X="--steps 20 --warmup 1 --no-extras --cpu-sample 0 --fastq-pairs 0 --index-cache /tmp/ixc"
timeout 300 python bench.py --steps 2 --warmup 1 --no-extras --cpu-sample 0 --fastq-pairs 0 --index-cache /tmp/ixc > /dev/null 2>&1
for i in 1 2 3; do
SQ_TIMING=1 timeout 300 python bench.py $X > $O/b_spin_$i.json 2> $O/b_spin_$i.err
SQ_TIMING=1 SQ_EM_SPIN=0 timeout 300 python bench.py $X > $O/b_sync_$i.json 2> $O/b_sync_$i.err
done
timeout 600 python -m pytest tests/test_em.py tests/test_scale_gpu.py -m gpu -x -q > $O/pytest_em.log 2>&1; tail -2 $O/pytest_em.log
echo done
