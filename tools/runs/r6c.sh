#!/bin/bash
# round 6, call C: the three-launch EM iteration — EM tests, then the c2 job without extras + kernel trace
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r6c; mkdir -p $O; cd $R
python -c "import torch" > /dev/null 2>&1
timeout -k 5 900 python -m pytest tests/test_em.py tests/test_vbem_pin.py tests/test_em_pin.py tests/test_bootstrap_pin.py tests/test_bias_gpu.py tests/test_scale_gpu.py tests/test_c1.py -m gpu -x -q > $O/gputests.txt 2>&1; tail -15 $O/gputests.txt
cd /tmp
timeout -k 5 400 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python $R/bench.py --steps 20 --warmup 5 --no-extras --cpu-sample 1000000 --fastq-pairs 0 --index-cache /tmp/ixc > $O/bench_kt.json 2> $O/bench_kt.err
db=$(find $O/kt -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/kstats.py $db "" 60 > $O/kernel_stats.txt; rm -rf $O/kt
grep -E "k_class|k_l1|k_fin|k_top|k_close|k_leaf" $O/kernel_stats.txt
python - <<PY
import json
d = json.loads(open("$O/bench_kt.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["breakdown"]["map_eq_s"], d["breakdown"]["em_call_s"], d["breakdown"]["em_iters"], d["em"]["ms_per_iter"], d["parity_check"]["equal"] if d.get("parity_check") else None, (d.get("parity_check") or {}).get("checks"))
PY
echo done
