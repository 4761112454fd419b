#!/bin/bash
# round 4, call B: full GPU test suite (3-launch EM iteration, B4 files), bench with 3 vs 5 launches per EM iteration
cd /root/repo; mkdir -p gpurun_out/r4b; O=gpurun_out/r4b
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1
X="--steps 20 --warmup 1 --no-extras --cpu-sample 0 --fastq-pairs 0 --index-cache /tmp/ixc"
timeout 300 python bench.py $X > $O/b_em3.json 2> $O/b_em3.err
SQ_EM_LAUNCHES=5 timeout 300 python bench.py $X > $O/b_em5.json 2> $O/b_em5.err
echo done
