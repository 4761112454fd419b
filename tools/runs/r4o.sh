#!/bin/bash
# round 4, call O: the device reader alone under its knobs; bench.py prints exactly one line on stdout
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
cd $R; mkdir -p gpurun_out/r4o; O=$R/gpurun_out/r4o
timeout 600 python tools/dev_reader_bench.py 20000000 2 > $O/dev_reader_2.txt 2>&1
timeout 600 python tools/dev_reader_bench.py 20000000 56 > $O/dev_reader_56.txt 2>&1
timeout 600 python bench.py --steps 2 --warmup 1 --cpu-sample 0 --fastq-pairs 0 --spread-pairs 2000000 > $O/b_stdout.json 2> $O/b_stdout.err
wc -l $O/b_stdout.json
echo done
