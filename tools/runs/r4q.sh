#!/bin/bash
# round 4, call Q: the whole GPU suite on the final sources, the bench lines that changed since call N (from_fastq through the ring reader, c3 without a
# foreign traffic figure), and the host reader's scaling with threads on compressed input
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r4q; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1
tail -3 $O/pytest_gpu.log
timeout 900 python bench.py --index-cache /tmp/ixc > $O/bench_c2_default.json 2> $O/bench_c2_default.err
wc -l $O/bench_c2_default.json
timeout 400 python bench.py --workload c3 --no-extras --cpu-sample 0 --fastq-pairs 0 --index-cache /tmp/ixc > $O/bench_c3_n1.json 2> $O/bench_c3_n1.err
for t in 32 64 128; do SQ_READER_DEVICE=0 timeout 300 python tools/reader_bench.py 3000000 $t > $O/reader_bench_t$t.txt 2>&1; done
echo done
