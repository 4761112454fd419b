#!/bin/bash
# round 6, call R: the inflater with up to three literals per table entry; peek width as built (LIT_BITS): tests, then the BGZF kernel alone and the gzip decoder alone
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r6r; mkdir -p $O; cd $R
python -c "import torch" > /dev/null 2>&1
grep -n "constexpr int LIT_BITS" salmon_amd/csrc/hip/inflate_core.h
timeout -k 5 900 python -m pytest tests/test_inflate.py tests/test_gzip_dev.py -x -q 2>&1 | tail -3
cd /tmp
timeout -k 5 400 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python $R/tools/inflate_bench.py > $O/ib.txt 2>&1
db=$(find $O/kt -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/kstats.py $db "" 6 > $O/ks.txt; rm -rf $O/kt; grep -v "^W2026" $O/ib.txt | tail -6 | cut -c1-200; head -4 $O/ks.txt | cut -c1-200
timeout -k 5 400 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python $R/tools/gzip_bench.py 400 > $O/gb.txt 2>&1
db=$(find $O/kt -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/kstats.py $db "" 6 > $O/ks2.txt; rm -rf $O/kt; head -5 $O/ks2.txt | cut -c1-200
echo done
