#!/bin/bash
# round 6, call D: EM block-plan sweep
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r6d; mkdir -p $O; cd $R
python -c "import torch" > /dev/null 2>&1
timeout -k 5 900 python tools/em_sweep.py 8 > $O/em_sweep.txt 2> $O/em_sweep.err; cat $O/em_sweep.txt; tail -5 $O/em_sweep.err
echo done
