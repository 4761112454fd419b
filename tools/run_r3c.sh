#!/bin/bash
# round 3: bitonic MEM sort for the wave-per-end classes (c4), seed speculation depth 1 vs 2 on c2
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r3c; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_map_gpu.py tests/test_golden.py -m gpu -q --timeout 300 > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
SQ_SEED_SPEC=1 timeout 400 python bench.py --steps 6 --warmup 1 --cpu-sample 0 --fastq-pairs 0 > $O/c2_spec1.json 2> $O/c2_spec1.err
timeout 900 python bench.py --workload c4 --steps 6 --warmup 1 > $O/bench_c4.json 2> $O/bench_c4.err
