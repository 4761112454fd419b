#!/bin/bash
# round 6, call ZI: the large ends' sort as a segmented sort (a block per end), the DP and the acceptance over LDS tiles (k_lg_dp2 / k_lg_accept2) against the whole-buffer
# sort and the thread-per-cluster kernels: mapping tests, then configs[3] at full size both ways (same index, same session), kernel trace of the new one
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r6zi; mkdir -p $O; cd $R
python -c "import torch" > /dev/null 2>&1
timeout -k 5 900 python -m pytest tests/test_map_gpu.py -m gpu -x -q > $O/gputests_map.txt 2>&1; grep -E "passed|failed|error" $O/gputests_map.txt | tail -3
C4="--workload c4 --genome-gnt 3.1 --warmup 1 --no-extras --fastq-pairs 0 --index-cache /tmp/ixc4"
show() { python - <<PY
import json
try:
    d = json.loads(open("$O/$1.json").read().strip().splitlines()[-1])
    print("$1", d["value"], d["ms_per_step"], d["breakdown"]["map_eq_s"], (d.get("parity_check") or {}).get("equal"), {k: v["avg_ms"] for k, v in d["stages"].items() if k in ("k_mems", "large_ends", "k_seed", "k_score")})
except Exception as e: print("$1 failed", e); print(open("$O/$1.err").read()[-800:])
PY
}
timeout 1500 python bench.py $C4 --steps 5 --cpu-sample 200000 > $O/bench_c4_new.json 2> $O/bench_c4_new.err; show bench_c4_new
SQ_LG_OLD=1 SQ_LG_GLOBAL_SORT=1 timeout 1500 python bench.py $C4 --steps 5 --cpu-sample 0 > $O/bench_c4_old.json 2> $O/bench_c4_old.err; show bench_c4_old
SQ_LG_GLOBAL_SORT=1 timeout 1500 python bench.py $C4 --steps 5 --cpu-sample 0 > $O/bench_c4_newdp_oldsort.json 2> $O/bench_c4_newdp_oldsort.err; show bench_c4_newdp_oldsort
cd /tmp
timeout -k 5 900 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python $R/bench.py $C4 --steps 3 --cpu-sample 0 > $O/kt_c4.json 2> $O/kt_c4.err
db=$(find $O/kt -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/kstats.py $db "" 70 > $O/kernel_stats_c4_full.txt; rm -rf $O/kt
head -34 $O/kernel_stats_c4_full.txt | cut -c1-170
echo done
