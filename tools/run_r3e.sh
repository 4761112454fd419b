#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r3e; mkdir -p $O
cd $R
timeout 600 python tools/reader_bench.py 4000000 > $O/reader_bench.txt 2>&1
SQ_TIMING=1 timeout 300 python - > $O/pgz_timing.txt 2>&1 <<'PY'
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import reader_bench as rb
print(rb.drain("/dev/shm/sq_reader_bench/r_1.fq.gz", "/dev/shm/sq_reader_bench/r_2.fq.gz", 1000000))
PY
