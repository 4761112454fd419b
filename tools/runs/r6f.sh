#!/bin/bash
# round 6, call F: is the late first synchronisation after a host-only pause a property of the platform?  (tools/stall_probe.py: plain torch)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r6f; mkdir -p $O; cd $R
timeout -k 5 600 python tools/stall_probe.py > $O/stall_probe.txt 2>&1; cat $O/stall_probe.txt
echo done
