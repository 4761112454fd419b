#!/bin/bash
# round 5, call N: the CU partition again, now that the mapping stream is faster — SQ_EQ_CUS = 32 / 48 / 64 on the driver's workload (10 steps)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r5n; mkdir -p $O; cd $R
python -c "import torch" > /dev/null 2>&1
Q="--steps 10 --warmup 2 --no-extras --cpu-sample 0 --fastq-pairs 0 --index-cache /tmp/ixc"
for cus in 64 32 48 64 32; do
  SQ_EQ_CUS=$cus timeout 400 python bench.py $Q > $O/b_$cus.json 2> $O/b_$cus.err
  python - <<PY
import json; d = json.loads(open("$O/b_$cus.json").read().strip().splitlines()[-1]); b = d["breakdown"]
print("SQ_EQ_CUS=$cus value %.1f map_eq_s %.4f tail %.4f k_seed %.3f ms eq_mini %.1f ms" % (d["value"], b["map_eq_s"], b["tail_s(eq_export+merge+normalize+EM)"], d["stages"]["k_seed"]["avg_ms"], d["stages"]["eq_mini_batches"]["ms_total"]))
PY
done 2>&1 | tee $O/summary.txt
echo done
