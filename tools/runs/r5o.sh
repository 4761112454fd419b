#!/bin/bash
# round 5, call O: kernel trace of the from-FASTQ legs (plain, gzip, BGZF with the members inflated on the device) — what the device spends on k_bgzf_inflate inside a real job
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r5o; mkdir -p $O; cd /tmp
python -c "import torch" > /dev/null 2>&1
timeout -k 5 500 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python $R/bench.py --steps 2 --warmup 1 --no-extras --cpu-sample 0 --index-cache /tmp/ixc > $O/bench.json 2> $O/bench.err
db=$(find $O/kt -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/kstats.py $db "" 40 > $O/kernel_stats_fastq_legs.txt; rm -rf $O/kt
head -30 $O/kernel_stats_fastq_legs.txt
python - <<PY
import json; d = json.loads(open("$O/bench.json").read().strip().splitlines()[-1]); print({k: (v.get("value") if isinstance(v, dict) else v) for k, v in d["from_fastq"].items() if k in ("plain", "gzip", "bgzf", "compressed_error")})
PY
echo done
