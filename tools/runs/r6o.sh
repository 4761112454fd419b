#!/bin/bash
# round 6, call O: the whole GPU suite on the current sources + the driver's bench command
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r6o; mkdir -p $O; cd $R
python -c "import torch" > /dev/null 2>&1
timeout -k 5 1500 python -m pytest tests -m gpu -x -q > $O/gputests.txt 2>&1; grep -E "passed|failed|error" $O/gputests.txt | tail -3
timeout -k 5 900 python bench.py --gpus 1 --steps 20 --warmup 5 --index-cache /tmp/ixc > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err
python - <<PY
import json
d = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["breakdown"]["map_eq_s"], d["breakdown"]["em_call_s"], d["breakdown"]["em_iters"], d["em"]["ms_per_iter"], d["breakdown"]["kfd_queues_evicted_ms_in_timed_region"])
print({k: d["roofline"].get(k) for k in ("kernel", "frac", "achieved", "avg_launch_ms", "traffic")}, d["parity_check"]["equal"] if d.get("parity_check") else None)
print({k: (v.get("value") if isinstance(v, dict) else v) for k, v in (d.get("from_fastq") or {}).items() if k in ("plain", "gzip", "bgzf", "compressed_error")})
PY
echo done
