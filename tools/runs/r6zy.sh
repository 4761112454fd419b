#!/bin/bash
# round 6, call ZY: the from-files legs on the metric's 100 M pairs once more, on the final tree
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r6zy; mkdir -p $O; cd $R
python -c "import torch" > /dev/null 2>&1
timeout -k 5 1500 python bench.py --steps 4 --warmup 1 --no-extras --cpu-sample 0 --fastq-pairs 100000000 --fastq-gz-pairs 100000000 --index-cache /tmp/ixc > $O/bench_fastq100m.json 2> $O/bench_fastq100m.err; tail -c 300 $O/bench_fastq100m.err
python - <<PY
import json
try:
    d = json.loads(open("$O/bench_fastq100m.json").read().strip().splitlines()[-1]); f = d.get("from_fastq") or {}
    print(f.get("input")); print({k: v for k, v in f.items() if k in ("plain", "gzip", "bgzf", "compressed_error")})
except Exception as e: print("fastq100m failed", e)
PY
echo done
