#!/bin/bash
# round 6, call ZS: batches in flight on the mapping lanes (sq_map_submit / sq_map_wait) at 5 M and 2.5 M pairs per call
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r6zs; mkdir -p $O; cd $R
python -c "import torch" > /dev/null 2>&1
run() { local lab=$1; shift
  timeout -k 5 500 python bench.py --steps 20 --warmup 5 --no-extras --cpu-sample 0 --fastq-pairs 0 --index-cache /tmp/ixc "$@" > $O/b_$lab.json 2> $O/b_$lab.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/b_$lab.json").read().strip().splitlines()[-1])
    print("$lab:", d["value"], "map_eq_s", d["breakdown"]["map_eq_s"], "tail", d["breakdown"]["tail_s(eq_export+merge+normalize+EM)"], "eqf", d["breakdown"]["eq_finish_s"], "em", d["breakdown"]["em_call_s"], d["breakdown"]["em_iters"], d["config"].get("pairs_per_step"), d["config"].get("job_pairs"))
except Exception as e:
    print("$lab failed", e); print(open("$O/b_$lab.err").read()[-600:])
PY
}
run l1_b5m --batch 5000000 --sub 1 --lanes 1
run l2_b5m --batch 5000000 --sub 1 --lanes 2
run l3_b5m --batch 5000000 --sub 1 --lanes 3
run l2_b2p5m --batch 2500000 --sub 2 --lanes 2
run l1_b5m_again --batch 5000000 --sub 1 --lanes 1
run l2_b5m_again --batch 5000000 --sub 1 --lanes 2
echo done
