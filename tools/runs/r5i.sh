#!/bin/bash
# round 5, call I: BGZF inflated on the device — the kernel against zlib, the reader against the host reader, the from_fastq legs of the bench, then the whole suite
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r5i; mkdir -p $O; cd $R
python -c "import torch" > /dev/null 2>&1
timeout 300 python -m pytest tests/test_inflate.py -m gpu -x -q > $O/pytest_inflate.log 2>&1; tail -5 $O/pytest_inflate.log
timeout 900 python -m pytest tests/test_reader_gpu.py -m gpu -x -q > $O/pytest_reader.log 2>&1; tail -5 $O/pytest_reader.log
SQ_READER_STATS=1 timeout 900 python bench.py --steps 4 --warmup 1 --no-extras --cpu-sample 0 --index-cache /tmp/ixc > $O/bench_fastq.json 2> $O/bench_fastq.err; tail -c 1500 $O/bench_fastq.json; grep -a "sq_dev_reader" $O/bench_fastq.err | tail -12
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_inflate -- python $R/tools/inflate_bench.py > $O/inflate_bench.log 2>&1; cd $R; tail -5 $O/inflate_bench.log
db=$(find $O/prof_inflate -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/kstats.py $db "" 6 > $O/inflate_kernel_stats.txt; cat $O/inflate_kernel_stats.txt
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu_all.log 2>&1; tail -3 $O/pytest_gpu_all.log
echo done
