#!/bin/bash
# round 6, call M: the reader with ordinary gzip files inflated on the device: tests, then the from-FASTQ legs of the bench
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r6m; mkdir -p $O; cd $R
python -c "import torch" > /dev/null 2>&1
timeout -k 5 900 python -m pytest tests/test_reader_gpu.py tests/test_gzip_dev.py -x -q 2>&1 | tail -30 | cut -c1-600
SQ_READER_STATS=1 timeout -k 5 600 python bench.py --steps 2 --warmup 1 --no-extras --cpu-sample 0 --index-cache /tmp/ixc > $O/bench.json 2> $O/bench.err; grep "sq_dev_reader" $O/bench.err | cut -c1-400
python - <<PY
import json
d = json.loads(open("$O/bench.json").read().strip().splitlines()[-1]); print(json.dumps(d["from_fastq"])[:1500])
PY
echo done
