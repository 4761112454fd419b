#!/bin/bash
# round 6, call X: the burned-in chain of mini-batch groups on a stream of its own, one batch behind the static stage and the class table (ctx.h: stream_chain) —
# the GPU suite, then the bench with the chain in line (SQ_CHAIN_STREAM=0), trailing on the eq partition's 64 CUs, and with the partition cut in several ways
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r6x; mkdir -p $O; cd $R
python -c "import torch" > /dev/null 2>&1
timeout -k 5 1500 python -m pytest tests -m gpu -x -q > $O/gputests.txt 2>&1; grep -E "passed|failed|error" $O/gputests.txt | tail -3
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
run inline SQ_CHAIN_STREAM=0
run trail64 SQ_X=1
run chain32_eq32 SQ_CHAIN_CUS=32
run chain16_eq48 SQ_CHAIN_CUS=16
run eq96_chain32 SQ_EQ_CUS=96 SQ_CHAIN_CUS=32
run eq48_chain16 SQ_EQ_CUS=48 SQ_CHAIN_CUS=16
run eq32_trail SQ_EQ_CUS=32
run trail64b SQ_X=1
run inlineb SQ_CHAIN_STREAM=0
echo done
