#!/bin/bash
# round 4, call P: the device reader on a ring of page-locked pieces: parity with the host reader, throughput alone and end to end
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
cd $R; mkdir -p gpurun_out/r4p; O=$R/gpurun_out/r4p
timeout 300 python -m pytest tests/test_reader_gpu.py -m gpu -x -q > $O/pytest_reader.log 2>&1
tail -3 $O/pytest_reader.log
timeout 600 python tools/dev_reader_bench.py 20000000 2 > $O/dev_reader_2.txt 2>&1
SQ_READER_STATS=1 timeout 600 python bench.py --steps 4 --warmup 1 --no-extras --cpu-sample 0 --fastq-pairs 20000000 --index-cache /tmp/ixc > $O/b_fastq.json 2> $O/b_fastq.err
echo done
