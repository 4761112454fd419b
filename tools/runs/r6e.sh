#!/bin/bash
# round 6, call E: EM tests + the c2 job (no extras) twice, plain and with SQ_TIMING, on the three-launch EM with 1024-entry blocks and the look-ahead loop
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r6e; mkdir -p $O; cd $R
python -c "import torch" > /dev/null 2>&1

for i in 1 2 3; do
SQ_TIMING=1 timeout -k 5 400 python bench.py --steps 20 --warmup 5 --no-extras --cpu-sample 1000000 --fastq-pairs 0 --index-cache /tmp/ixc > $O/bench_$i.json 2> $O/bench_$i.err
python - <<PY
import json
d = json.loads(open("$O/bench_$i.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["breakdown"]["map_eq_s"], d["breakdown"]["eq_finish_s"], d["breakdown"]["normalize_alphas_s"], d["breakdown"]["em_call_s"], d["breakdown"]["em_iters"], d["em"]["ms_per_iter"], d["breakdown"]["host_cpu_in_timed_region"])
PY
grep -E "^\[em" $O/bench_$i.err | tail -40
done
echo done
