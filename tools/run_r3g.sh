#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r3g; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_map_gpu.py tests/test_scale_gpu.py tests/test_exhaustive.py tests/test_poison_gpu.py -m gpu -q --timeout 300 > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
timeout 400 python bench.py --steps 6 --warmup 1 --cpu-sample 200000 --fastq-pairs 0 > $O/c2.json 2> $O/c2.err
SQ_MEMS_G16=1 timeout 400 python bench.py --steps 6 --warmup 1 --cpu-sample 0 --fastq-pairs 0 > $O/c2_g16.json 2> $O/c2_g16.err
