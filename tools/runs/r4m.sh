#!/bin/bash
# round 4, call M: flat chaining of the large ends: parity tests (small), then configs[3] at full size (before: profiles/r04_bench_c4_full.json)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
cd $R; mkdir -p gpurun_out/r4m; O=$R/gpurun_out/r4m
timeout 900 python -m pytest tests/test_map_gpu.py tests/test_long_reads.py -m gpu -x -q > $O/pytest_map.log 2>&1
tail -3 $O/pytest_map.log
if grep -q failed $O/pytest_map.log; then echo "parity failed: stopping"; exit 0; fi
C4="--workload c4 --genome-gnt 3.1 --steps 5 --warmup 1 --no-extras --cpu-sample 200000 --fastq-pairs 0"
timeout 1500 python bench.py $C4 > $O/b_c4_flat.json 2> $O/b_c4_flat.err
echo done
