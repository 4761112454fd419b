#!/bin/bash
# round 6, call Y: the trailing chain with a smaller eq partition shared by the chain and the static stage / class table (call X: on 64 CUs the trailing chain changes nothing,
# the mapping stream on its 192 CUs is what a step waits for)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r6y; mkdir -p $O; cd $R
python -c "import torch" > /dev/null 2>&1
run() {  # label, env...
  local lab=$1; shift
  env "$@" timeout -k 5 400 python bench.py --steps 10 --warmup 2 --no-extras --cpu-sample 0 --fastq-pairs 0 --index-cache /tmp/ixc > $O/b_$lab.json 2> $O/b_$lab.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/b_$lab.json").read().strip().splitlines()[-1])
    print("$lab:", d["value"], "map_eq_s", d["breakdown"]["map_eq_s"], "tail", d["breakdown"]["tail_s(eq_export+merge+normalize+EM)"], "eqf", d["breakdown"]["eq_finish_s"], {k: v["avg_ms"] for k, v in d["stages"].items() if k in ("k_seed", "k_mems", "k_score", "eq_static", "eq_table", "eq_flags_scan")}, "mini", d["stages"]["eq_mini_batches"]["ms_total"])
except Exception as e:
    print("$lab: failed", e); print(open("$O/b_$lab.err").read()[-600:])
PY
}
run trail64 SQ_X=1
run trail48 SQ_EQ_CUS=48
run trail56 SQ_EQ_CUS=56
run trail40 SQ_EQ_CUS=40
run inline48 SQ_EQ_CUS=48 SQ_CHAIN_STREAM=0
run inline64 SQ_CHAIN_STREAM=0
echo done
