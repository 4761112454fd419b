#!/bin/bash
# round 6, call ZL: the tile's clusters taken largest first (a wave's 64 clusters of about one size), tiles of 2 048 records and 256 threads: mapping tests,
# configs[3] at full size with parity, kernel trace
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r6zl; mkdir -p $O; cd $R
python -c "import torch" > /dev/null 2>&1
timeout -k 5 900 python -m pytest tests/test_map_gpu.py -m gpu -x -q > $O/gputests_map.txt 2>&1; grep -E "passed|failed|error" $O/gputests_map.txt | tail -3
C4="--workload c4 --genome-gnt 3.1 --warmup 1 --no-extras --fastq-pairs 0 --index-cache /tmp/ixc4"
timeout 1500 python bench.py $C4 --steps 5 --cpu-sample 200000 > $O/bench_c4_full.json 2> $O/bench_c4_full.err
python - <<PY
import json
try:
    d = json.loads(open("$O/bench_c4_full.json").read().strip().splitlines()[-1])
    print("c4", d["value"], d["ms_per_step"], d["breakdown"]["map_eq_s"], (d.get("parity_check") or {}).get("equal"), {k: v["avg_ms"] for k, v in d["stages"].items()})
except Exception as e: print("c4 failed", e); print(open("$O/bench_c4_full.err").read()[-800:])
PY
cd /tmp
timeout -k 5 900 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python $R/bench.py $C4 --steps 3 --cpu-sample 0 > $O/kt_c4.json 2> $O/kt_c4.err
db=$(find $O/kt -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/kstats.py $db "" 70 > $O/kernel_stats_c4_full.txt; rm -rf $O/kt
head -34 $O/kernel_stats_c4_full.txt | cut -c1-170
echo done
