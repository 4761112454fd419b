#!/bin/bash
# round 3, second half: --posBias + the condition-free k_dp on the GPU, then kernel stats and a lanes x seed-grid sweep
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r3b; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_bias_gpu.py tests/test_map_gpu.py tests/test_golden.py tests/test_exhaustive.py tests/test_scale_gpu.py tests/test_bench_contract.py -m gpu -q --timeout 300 > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
cd $R
for cfg in "1 6" "2 6" "2 4" "2 3"; do set -- $cfg
  SQ_SEED_BPC=$2 timeout 400 python bench.py --steps 6 --warmup 1 --cpu-sample 0 --fastq-pairs 0 --lanes $1 > $O/sweep_l$1_b$2.json 2> $O/sweep_l$1_b$2.err
done
