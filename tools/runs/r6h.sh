#!/bin/bash
# round 6, call H2: is the late first synchronisation a queue eviction?  amdkfd's evicted_ms around the timed region
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r6h; mkdir -p $O; cd $R
python -c "import torch" > /dev/null 2>&1
ls /sys/class/kfd/kfd/proc/ 2>&1 | head -3; ls /sys/class/kfd/kfd/proc/*/ 2>&1 | head -20
for v in 1 2 3; do
SQ_TIMING=1 timeout -k 5 400 python bench.py --steps 20 --warmup 5 --no-extras --cpu-sample 0 --fastq-pairs 0 --index-cache /tmp/ixc > $O/bench_$v.json 2> $O/bench_$v.err
echo "$(grep 'upload drained' $O/bench_$v.err | tr '\n' ' ')"
python - <<PY
import json
d = json.loads(open("$O/bench_$v.json").read().strip().splitlines()[-1])
print(d["value"], d["breakdown"]["map_eq_s"], d["breakdown"]["em_call_s"], d["breakdown"]["kfd_queues_evicted_ms_in_timed_region"])
PY
done
echo done
