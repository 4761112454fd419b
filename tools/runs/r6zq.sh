#!/bin/bash
# round 6, call ZQ: the eq wait in front of k_select instead of k_score — the measurement set again on the round's FINAL kernels — the whole GPU suite, the driver's bench command, a kernel trace of it, the FETCH_SIZE / WRITE_SIZE passes
# (separate; kernel-trace only), configs[3] at full size (bench + kernel trace come from call ZJ on the same large-end kernels; here the bench line again on the final tree)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r6zq; mkdir -p $O; cd $R
python -c "import torch" > /dev/null 2>&1
timeout -k 5 1500 python -m pytest tests -m gpu -x -q > $O/gputests.txt 2>&1; grep -E "passed|failed|error" $O/gputests.txt | tail -3
timeout -k 5 900 python bench.py --gpus 1 --steps 20 --warmup 5 --index-cache /tmp/ixc > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err
cd /tmp
timeout -k 5 400 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python $R/bench.py --steps 20 --warmup 5 --no-extras --cpu-sample 0 --fastq-pairs 0 --index-cache /tmp/ixc > $O/bench_kt.json 2> $O/bench_kt.err
db=$(find $O/kt -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/kstats.py $db "" 70 > $O/kernel_stats.txt; rm -rf $O/kt
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 5 300 rocprofv3 --kernel-trace --pmc $c -d $O/pmc_$c -o pmc --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-extras --cpu-sample 0 --fastq-pairs 0 --index-cache /tmp/ixc > /dev/null 2> $O/pmc_$c.err
  python $R/tools/pmc_summary.py $O/pmc_$c 60 $O/pmc_$c.json > $O/pmc_$c.txt; rm -rf $O/pmc_$c
done
head -8 $O/pmc_FETCH_SIZE.txt
cd $R
python - <<PY
import json
d = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["breakdown"]["map_eq_s"], d["breakdown"]["em_call_s"], d["breakdown"]["em_iters"], d["em"]["ms_per_iter"], d["breakdown"]["index_build_s"])
print({k: d["roofline"].get(k) for k in ("kernel", "frac", "achieved", "avg_launch_ms", "traffic")}, d["parity_check"]["equal"] if d.get("parity_check") else None, (d.get("parity_check") or {}).get("checks"))
print({k: (v.get("value") if isinstance(v, dict) else v) for k, v in (d.get("from_fastq") or {}).items() if k in ("plain", "gzip", "bgzf", "compressed_error")})
print({k: v["avg_ms"] for k, v in d["stages"].items()})
print("c2s", (d.get("c2s") or {}).get("value"), "jobs", {k: (v.get("value") if isinstance(v, dict) else v) for k, v in (d.get("jobs") or {}).items()})
PY
C4="--workload c4 --genome-gnt 3.1 --warmup 1 --no-extras --fastq-pairs 0 --index-cache /tmp/ixc4"
SQ_TIMING=1 timeout 1500 python bench.py $C4 --steps 5 --cpu-sample 200000 > $O/bench_c4_full.json 2> $O/bench_c4_full.err; grep "sq-timing\] index" $O/bench_c4_full.err | tee $O/index_phases_c4.txt | tail -3
python - <<PY
import json
try:
    d = json.loads(open("$O/bench_c4_full.json").read().strip().splitlines()[-1])
    print("c4_full", d["value"], d["ms_per_step"], d["breakdown"]["map_eq_s"], (d.get("parity_check") or {}).get("equal"), d["breakdown"]["index_build_s"])
except Exception as e: print("c4 failed", e)
PY
timeout 400 python bench.py --workload c3 --no-extras --cpu-sample 0 --fastq-pairs 0 --index-cache /tmp/ixc > $O/bench_c3_n1.json 2> $O/bench_c3_n1.err
timeout 400 python bench.py --workload c5 --no-extras --cpu-sample 200000 --fastq-pairs 0 --index-cache /tmp/ixc > $O/bench_c5.json 2> $O/bench_c5.err
python - <<PY
import json
for f in ("bench_c3_n1", "bench_c5"):
    try:
        d = json.loads(open("$O/%s.json" % f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"], d["breakdown"]["map_eq_s"], (d.get("parity_check") or {}).get("equal"))
    except Exception as e: print(f, "failed", e)
PY
echo done
