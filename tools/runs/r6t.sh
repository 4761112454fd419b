#!/bin/bash
# round 6, call T: the burned-in mini-batch chain as one launch per batch (k_chain) — the whole GPU suite on it, then the chain as launch pairs (SQ_CHAIN=0) against
# the one launch, and the CU partition again (the chain no longer needs a quarter of the chip to itself)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r6t; mkdir -p $O; cd $R
python -c "import torch" > /dev/null 2>&1
timeout -k 5 1500 python -m pytest tests -m gpu -x -q > $O/gputests.txt 2>&1; grep -E "passed|failed|error" $O/gputests.txt | tail -3
run() {  # label, env...
  local lab=$1; shift
  env "$@" timeout -k 5 400 python bench.py --steps 10 --warmup 2 --no-extras --cpu-sample 0 --fastq-pairs 0 --index-cache /tmp/ixc > $O/b_$lab.json 2> $O/b_$lab.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/b_$lab.json").read().strip().splitlines()[-1])
    print("$lab:", d["value"], "map_eq_s", d["breakdown"]["map_eq_s"], "tail", d["breakdown"]["tail_s(eq_export+merge+normalize+EM)"], {k: v["avg_ms"] for k, v in d["stages"].items() if k in ("k_seed", "k_mems", "k_score", "eq_static", "eq_table", "eq_flags_scan")}, "mini", d["stages"]["eq_mini_batches"]["ms_total"], "parity", (d.get("parity_check") or {}).get("equal"))
except Exception as e:
    print("$lab: failed", e); print(open("$O/b_$lab.err").read()[-600:])
PY
}
run pairs SQ_CHAIN=0
run chain SQ_CHAIN=1
run chain_cu48 SQ_CHAIN=1 SQ_EQ_CUS=48
run chain_cu32 SQ_CHAIN=1 SQ_EQ_CUS=32
run chain_cu0 SQ_CHAIN=1 SQ_EQ_CUS=0
run chain_w96 SQ_CHAIN=1 SQ_CHAIN_WGS=96
run chain2 SQ_CHAIN=1
timeout -k 5 900 python bench.py --gpus 1 --steps 20 --warmup 5 --index-cache /tmp/ixc > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err
python - <<PY
import json
d = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["breakdown"]["map_eq_s"], d["breakdown"]["em_call_s"], d["breakdown"]["em_iters"], d["em"]["ms_per_iter"])
print({k: d["roofline"].get(k) for k in ("kernel", "frac", "achieved", "avg_launch_ms", "traffic")}, d["parity_check"])
PY
echo done
