#!/bin/bash
# round 6, call N: the device gzip decoder alone (tools/gzip_bench.py) + a kernel trace of it
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r6n; mkdir -p $O; cd $R
python -c "import torch" > /dev/null 2>&1
timeout -k 5 600 python tools/gzip_bench.py 400 2>&1 | tail -12
cd /tmp
timeout -k 5 400 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python $R/tools/gzip_bench.py 400 > $O/gzb.txt 2>&1
db=$(find $O/kt -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/kstats.py $db "" 20 > $O/kernel_stats_gzip.txt; rm -rf $O/kt
head -12 $O/kernel_stats_gzip.txt
echo done
