#!/bin/bash
# round 6, call Q: FETCH_SIZE calibrated on known byte counts in k_seed2's access patterns (tools/fetch_calib.hip), then the measurement set on the round's kernels:
# the whole GPU suite, the driver's bench command, a kernel trace of it, the FETCH_SIZE / WRITE_SIZE passes (separate; kernel-trace only)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r6q; mkdir -p $O; cd /tmp
python -c "import torch" > /dev/null 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 5 200 rocprofv3 --kernel-trace --pmc $c -d $O/cal_$c -o pmc --output-format csv -- $R/tools/_build/fetch_calib > $O/fetch_calib_asked.txt 2> $O/cal_$c.err
  python $R/tools/pmc_summary.py $O/cal_$c 10 > $O/fetch_calib_$c.txt; rm -rf $O/cal_$c
done
cat $O/fetch_calib_asked.txt $O/fetch_calib_FETCH_SIZE.txt $O/fetch_calib_WRITE_SIZE.txt
cd $R
timeout -k 5 1500 python -m pytest tests -m gpu -x -q > $O/gputests.txt 2>&1; grep -E "passed|failed|error" $O/gputests.txt | tail -3
timeout -k 5 900 python bench.py --gpus 1 --steps 20 --warmup 5 --index-cache /tmp/ixc > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err
cd /tmp
timeout -k 5 400 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python $R/bench.py --steps 20 --warmup 5 --no-extras --cpu-sample 0 --fastq-pairs 0 --index-cache /tmp/ixc > $O/bench_kt.json 2> $O/bench_kt.err
db=$(find $O/kt -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/kstats.py $db "" 70 > $O/kernel_stats.txt; rm -rf $O/kt
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 5 300 rocprofv3 --kernel-trace --pmc $c -d $O/pmc_$c -o pmc --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-extras --cpu-sample 0 --fastq-pairs 0 --index-cache /tmp/ixc > /dev/null 2> $O/pmc_$c.err
  python $R/tools/pmc_summary.py $O/pmc_$c 60 $O/pmc_$c.json > $O/pmc_$c.txt; rm -rf $O/pmc_$c
done
head -14 $O/pmc_FETCH_SIZE.txt
python - <<PY
import json
d = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["breakdown"]["map_eq_s"], d["breakdown"]["em_call_s"], d["breakdown"]["em_iters"], d["em"]["ms_per_iter"])
print({k: d["roofline"].get(k) for k in ("kernel", "frac", "achieved", "avg_launch_ms", "traffic")}, d["parity_check"]["equal"] if d.get("parity_check") else None)
print({k: (v.get("value") if isinstance(v, dict) else v) for k, v in (d.get("from_fastq") or {}).items() if k in ("plain", "gzip", "bgzf", "compressed_error")})
PY
echo done
