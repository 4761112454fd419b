#!/bin/bash
# round 6, call J: whole GPU suite + the driver's bench command + kernel trace on the three-launch EM
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r6j; mkdir -p $O; cd $R
python -c "import torch" > /dev/null 2>&1
timeout -k 5 1200 python -m pytest tests -m gpu -x -q > $O/gputests.txt 2>&1; tail -4 $O/gputests.txt
timeout -k 5 600 python bench.py --gpus 1 --steps 20 --warmup 5 --index-cache /tmp/ixc > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err
cd /tmp
timeout -k 5 400 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python $R/bench.py --steps 20 --warmup 5 --no-extras --cpu-sample 0 --fastq-pairs 0 --index-cache /tmp/ixc > $O/bench_kt.json 2> $O/bench_kt.err
db=$(find $O/kt -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/kstats.py $db "" 60 > $O/kernel_stats.txt; rm -rf $O/kt
head -24 $O/kernel_stats.txt
python - <<PY
import json
d = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["breakdown"]["map_eq_s"], d["breakdown"]["em_call_s"], d["breakdown"]["em_iters"], d["em"]["ms_per_iter"], d["breakdown"]["kfd_queues_evicted_ms_in_timed_region"])
print({k: d["roofline"].get(k) for k in ("kernel", "frac", "achieved", "avg_launch_ms", "traffic")}, d["parity_check"]["equal"] if d.get("parity_check") else None)
print({k: (v.get("value") if isinstance(v, dict) else v) for k, v in (d.get("from_fastq") or {}).items() if k in ("plain", "gzip", "bgzf", "compressed_error")})
print(d["jobs"], d["c2s"]["value"] if d.get("c2s") else None)
PY
echo done
