#!/bin/bash
# round 5, call F (E again after the fixes): rocPRIM in place of hipCUB, BGZF members inflated straight into the ring, the 1e-4 test — GPU suite, then the from-FASTQ legs on 20 M pairs
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r5f; mkdir -p $O; cd $R
python -c "import torch" > /dev/null 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu_all.log 2>&1; tail -3 $O/pytest_gpu_all.log
SQ_READER_STATS=1 timeout 900 python bench.py --steps 4 --warmup 1 --no-extras --cpu-sample 0 --index-cache /tmp/ixc > $O/fastq.json 2> $O/fastq.err
python - <<'PY'
import json, os
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.getcwd()), "gpurun_out", "r5f")
try:
    d = json.loads(open(O + "/fastq.json").read().strip().splitlines()[-1]); f = d.get("from_fastq") or {}
    for k in ("plain", "gzip", "bgzf", "compressed_error"): print(k, f.get(k))
except Exception as e: print("fastq leg unreadable", e)
PY
grep "sq_dev_reader" $O/fastq.err | tail -8
echo done
