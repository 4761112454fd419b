#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r3d; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_map_gpu.py tests/test_scale_gpu.py tests/test_exhaustive.py -m gpu -q --timeout 300 > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
timeout 400 python bench.py --steps 6 --warmup 1 --cpu-sample 0 --fastq-pairs 0 > $O/c2.json 2> $O/c2.err
SQ_NO_UINFO=1 timeout 400 python bench.py --steps 6 --warmup 1 --cpu-sample 0 --fastq-pairs 0 > $O/c2_nouinfo.json 2> $O/c2_nouinfo.err
timeout 900 python bench.py --workload c4 --steps 6 --warmup 1 > $O/bench_c4.json 2> $O/bench_c4.err
