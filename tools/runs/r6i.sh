#!/bin/bash
# round 6, call I2: A/B in one call — the runtime's pinning of pageable copies (GPU_PINNED_MIN_XFER_SIZE unset / beyond any copy) against the late first synchronisation
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r6i; mkdir -p $O; cd $R
python -c "import torch" > /dev/null 2>&1
cat /proc/sys/kernel/numa_balancing 2>/dev/null; cat /sys/kernel/mm/transparent_hugepage/enabled 2>/dev/null
for v in unset 1048576 unset 1048576 unset 1048576; do
if [ $v = unset ]; then unset GPU_PINNED_MIN_XFER_SIZE; else export GPU_PINNED_MIN_XFER_SIZE=$v; fi
SQ_TIMING=1 timeout -k 5 400 python bench.py --steps 20 --warmup 5 --no-extras --cpu-sample 0 --fastq-pairs 0 --index-cache /tmp/ixc > $O/bench_$v.json 2> $O/bench_$v.err
echo "GPU_PINNED_MIN_XFER_SIZE=$v $(grep 'upload drained' $O/bench_$v.err | sed -n 2p)"
python - <<PY
import json
d = json.loads(open("$O/bench_$v.json").read().strip().splitlines()[-1])
print(d["value"], d["breakdown"]["map_eq_s"], d["breakdown"]["em_call_s"], d["breakdown"]["kfd_queues_evicted_ms_in_timed_region"], d["breakdown"]["read_gen_and_park_s"])
PY
done
echo done
