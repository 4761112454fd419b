#!/bin/bash
# round 6, call G: HIP + HSA + kernel + copy trace of the c2 job — the late first synchronisation of the EM set-up (tools/syncstall.py)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r6g; mkdir -p $O; cd /tmp
python -c "import torch" > /dev/null 2>&1
SQ_TIMING=1 timeout -k 5 600 rocprofv3 --hip-trace --hsa-trace --kernel-trace --memory-copy-trace -d $O/tr -o tr -- python $R/bench.py --steps 20 --warmup 5 --no-extras --cpu-sample 0 --fastq-pairs 0 --index-cache /tmp/ixc > $O/bench.json 2> $O/bench.err
grep "upload drained" $O/bench.err
db=$(find $O/tr -name "*.db" | head -1); ls -la $db
[ -n "$db" ] && python $R/tools/syncstall.py $db > $O/syncstall.txt 2>&1; rm -rf $O/tr
head -150 $O/syncstall.txt
echo done
