#!/bin/bash
# round 6, call ZW: the chain one batch behind on its own stream once more (patch re-applied on the final tree), with one and two calls in flight: the whole GPU suite with it, then the
# 100 M-pair job four ways
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r6zw; mkdir -p $O; cd $R
python -c "import torch" > /dev/null 2>&1
timeout -k 5 1500 python -m pytest tests -m gpu -x -q > $O/gputests.txt 2>&1; grep -E "passed|failed|error" $O/gputests.txt | tail -3
run() { local lab=$1; shift; local envs=$1; shift
  env $envs timeout -k 5 500 python bench.py --steps 20 --warmup 5 --no-extras --cpu-sample 0 --fastq-pairs 0 --index-cache /tmp/ixc "$@" > $O/b_$lab.json 2> $O/b_$lab.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/b_$lab.json").read().strip().splitlines()[-1])
    print("$lab:", d["value"], "map_eq_s", d["breakdown"]["map_eq_s"], "tail", d["breakdown"]["tail_s(eq_export+merge+normalize+EM)"], "eqf", d["breakdown"]["eq_finish_s"], {k: v["avg_ms"] for k, v in d["stages"].items() if k in ("k_seed", "k_select", "k_score", "eq_static", "eq_table")}, "mini", d["stages"]["eq_mini_batches"]["ms_total"])
except Exception as e:
    print("$lab failed", e); print(open("$O/b_$lab.err").read()[-600:])
PY
}
run inline_l1 SQ_CHAIN_STREAM=0 --lanes 1
run trail_l1 SQ_X=1 --lanes 1
run inline_l2 SQ_CHAIN_STREAM=0 --lanes 2
run trail_l2 SQ_X=1 --lanes 2
run inline_l1b SQ_CHAIN_STREAM=0 --lanes 1
run trail_l2b SQ_X=1 --lanes 2
echo done
