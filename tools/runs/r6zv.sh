#!/bin/bash
# round 6, call ZV: the driver's command once more on the committed tree (the line with the PMC traffic of the final sources attached)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r6zv; mkdir -p $O; cd $R
python -c "import torch" > /dev/null 2>&1
timeout -k 5 900 python bench.py --gpus 1 --steps 20 --warmup 5 --index-cache /tmp/ixc > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err
python - <<PY
import json
d = json.loads(open("$O/bench.json").read().strip().splitlines()[-1]); r = d["roofline"]
print(d["value"], d["ms_per_step"], d["breakdown"]["map_eq_s"], d["parity_check"]["equal"])
print({k: r.get(k) for k in ("kernel", "frac", "achieved", "avg_launch_ms", "traffic", "traffic_over_alg_bytes")}, r.get("valu_issue", {}).get("frac_of_issue_cycles"))
print({k: (v.get("value") if isinstance(v, dict) else v) for k, v in (d.get("from_fastq") or {}).items() if k in ("plain", "gzip", "bgzf")}, d["cpu_baseline"]["value"])
PY
echo done
