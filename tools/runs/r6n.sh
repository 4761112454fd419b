#!/bin/bash
# round 6, call N4: pipelined device gzip decoder: tests (decoder + reader), alone on both kinds of text, then the from-FASTQ legs
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r6n; mkdir -p $O; cd $R
python -c "import torch" > /dev/null 2>&1
timeout -k 5 900 python -m pytest tests/test_gzip_dev.py tests/test_reader_gpu.py -x -q 2>&1 | tail -12 | cut -c1-500
timeout -k 5 400 python tools/gzip_bench.py 800 const 2>&1 | grep -v "^W2026" | tail -2 | cut -c1-330
timeout -k 5 400 python tools/gzip_bench.py 400 2>&1 | grep -v "^W2026" | tail -2 | cut -c1-330
SQ_READER_STATS=1 timeout -k 5 600 python bench.py --steps 2 --warmup 1 --no-extras --cpu-sample 0 --index-cache /tmp/ixc > $O/bench.json 2> $O/bench.err; grep "sq_gzdev\|BGZF inflated" $O/bench.err | cut -c1-400
python - <<PY
import json
d = json.loads(open("$O/bench.json").read().strip().splitlines()[-1]); f = d["from_fastq"]
for k in ("plain", "gzip", "bgzf"): print(k, f.get(k))
PY
echo done
