#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r3h; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_sampling.py tests/test_dist_gpu.py tests/test_poison_gpu.py -m gpu -q --timeout 300 > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
timeout 600 python bench.py --workload c5 > $O/bench_c5.json 2> $O/bench_c5.err
