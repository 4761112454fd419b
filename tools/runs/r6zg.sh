#!/bin/bash
# round 6, call ZG: trailing chain without a CU partition, with the mapping stream over all CUs, with stream priorities
# the mapping stream on its 192 CUs is what a step waits for)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r6zg; mkdir -p $O; cd $R
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
run inline64 SQ_CHAIN_STREAM=0
run trail64 SQ_X=1
run trail_nopart SQ_EQ_CUS=0
run trail_nopart_prio SQ_EQ_CUS=0 SQ_STREAM_PRIO=1
run inline_nopart SQ_EQ_CUS=0 SQ_CHAIN_STREAM=0
run inline_nopart_prio SQ_EQ_CUS=0 SQ_CHAIN_STREAM=0 SQ_STREAM_PRIO=1
run trail_mapall64 SQ_MAP_ALL_CUS=1
run inline_mapall64 SQ_MAP_ALL_CUS=1 SQ_CHAIN_STREAM=0
run trail_mapall32 SQ_MAP_ALL_CUS=1 SQ_EQ_CUS=32
run trail_mapall128 SQ_MAP_ALL_CUS=1 SQ_EQ_CUS=128
echo done
