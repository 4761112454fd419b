#!/bin/bash
# round 6, call ZF: the whole GPU suite on the tree without the 257..1024-MEM class, then a kernel trace of configs[3] at full size (what the flat large-end passes are made of now)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r6zf; mkdir -p $O; cd $R
python -c "import torch" > /dev/null 2>&1
timeout -k 5 1500 python -m pytest tests -m gpu -x -q > $O/gputests.txt 2>&1; grep -E "passed|failed|error" $O/gputests.txt | tail -3
C4="--workload c4 --genome-gnt 3.1 --warmup 1 --no-extras --fastq-pairs 0 --index-cache /tmp/ixc4"
timeout 1500 python bench.py $C4 --steps 5 --cpu-sample 200000 > $O/bench_c4_full.json 2> $O/bench_c4_full.err
cd /tmp
timeout -k 5 900 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python $R/bench.py $C4 --steps 3 --cpu-sample 0 > $O/kt_c4.json 2> $O/kt_c4.err
db=$(find $O/kt -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/kstats.py $db "" 70 > $O/kernel_stats_c4_full.txt; rm -rf $O/kt
head -30 $O/kernel_stats_c4_full.txt | cut -c1-170
cd $R
python - <<PY
import json
d = json.loads(open("$O/bench_c4_full.json").read().strip().splitlines()[-1])
print("c4_full", d["value"], d["ms_per_step"], d["breakdown"]["map_eq_s"], (d.get("parity_check") or {}).get("equal"), d["breakdown"]["index_build_s"], {k: v["avg_ms"] for k, v in d["stages"].items()})
PY
echo done
