#!/bin/bash
# round 4, call G: new tests (long reads fix, shared-prefix dist), the default bench line with the new spread, c3 at N=1, reader thread scaling
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
cd $R; mkdir -p gpurun_out/r4g; O=$R/gpurun_out/r4g
timeout 900 python -m pytest tests/test_long_reads.py tests/test_dist_gpu.py tests/test_bench_contract.py -m gpu -q > $O/pytest_new.log 2>&1
timeout 900 python bench.py --steps 20 --warmup 2 --index-cache /tmp/ixc > $O/bench_c2.json 2> $O/bench_c2.err
timeout 600 python bench.py --workload c3 --steps 20 --warmup 1 --cpu-sample 0 --fastq-pairs 0 --index-cache /tmp/ixc > $O/bench_c3_n1.json 2> $O/bench_c3_n1.err
for t in 32 64 128; do timeout 300 python tools/reader_bench.py 8000000 $t > $O/reader_$t.txt 2>&1; done
echo done
