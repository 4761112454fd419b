#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r3j; mkdir -p $O
cd $R
timeout 400 python -m pytest tests/test_bench_contract.py -m gpu -q --timeout 300 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
SQ_MAP_ALL_CUS=1 timeout 400 python bench.py --steps 8 --warmup 1 --cpu-sample 0 --fastq-pairs 0 > $O/c2_allcus.json 2> $O/c2_allcus.err
timeout 400 python bench.py --steps 8 --warmup 1 --cpu-sample 0 --fastq-pairs 0 > $O/c2_base.json 2> $O/c2_base.err
SQ_MAP_ALL_CUS=1 SQ_SEED_BPC=5 timeout 400 python bench.py --steps 8 --warmup 1 --cpu-sample 0 --fastq-pairs 0 > $O/c2_allcus_b5.json 2> $O/c2_allcus_b5.err
