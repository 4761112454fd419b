#!/bin/bash
# round 6, call ZK: k_select's counters leave as per-block rows (k_select_stats adds them up), k_count_kmer_frags on a grid of 1 024 blocks, k_pack four characters at a time:
# the whole GPU suite, then c2 (10 steps, stage timers) and a kernel trace
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r6zk; mkdir -p $O; cd $R
python -c "import torch" > /dev/null 2>&1
timeout -k 5 1500 python -m pytest tests -m gpu -x -q > $O/gputests.txt 2>&1; grep -E "passed|failed|error" $O/gputests.txt | tail -3
for i in 1 2; do
timeout -k 5 400 python bench.py --steps 10 --warmup 2 --no-extras --cpu-sample 0 --fastq-pairs 0 --index-cache /tmp/ixc > $O/b_$i.json 2> $O/b_$i.err
python - <<PY
import json
try:
    d = json.loads(open("$O/b_$i.json").read().strip().splitlines()[-1])
    print("c2 run $i:", d["value"], "map_eq_s", d["breakdown"]["map_eq_s"], "tail", d["breakdown"]["tail_s(eq_export+merge+normalize+EM)"], {k: v["avg_ms"] for k, v in d["stages"].items()})
except Exception as e:
    print("failed", e); print(open("$O/b_$i.err").read()[-600:])
PY
done
cd /tmp
timeout -k 5 400 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python $R/bench.py --steps 10 --warmup 2 --no-extras --cpu-sample 0 --fastq-pairs 0 --index-cache /tmp/ixc > $O/bench_kt.json 2> $O/bench_kt.err
db=$(find $O/kt -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/kstats.py $db "" 70 > $O/kernel_stats.txt; rm -rf $O/kt
head -28 $O/kernel_stats.txt | cut -c1-170
echo done
