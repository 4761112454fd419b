#!/bin/bash
# the resident chain kernel (SQ_EQ_CHAIN=1): parity first (every command under its own timeout: a barrier that never opens must not hang the box), then the bench
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r3l; mkdir -p $O
cd $R
SQ_EQ_CHAIN=1 timeout -k 5 150 python -m pytest tests/test_map_gpu.py -m gpu -q --timeout 120 -k "burn_in or scale or stages" > $O/pytest1.log 2>&1; echo "pytest rc=$?" >> $O/pytest1.log
if grep -q "rc=0" $O/pytest1.log; then
  SQ_EQ_CHAIN=1 timeout -k 5 240 python -m pytest tests/test_scale_gpu.py tests/test_bias_gpu.py tests/test_poison_gpu.py -m gpu -q --timeout 200 > $O/pytest2.log 2>&1; echo "pytest rc=$?" >> $O/pytest2.log
  SQ_EQ_CHAIN=1 timeout -k 5 300 python bench.py --steps 8 --warmup 1 --cpu-sample 200000 --fastq-pairs 0 > $O/c2_chain.json 2> $O/c2_chain.err
fi
