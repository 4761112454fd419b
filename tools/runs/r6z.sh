#!/bin/bash
# round 6, call Z: k_seed2 with one load per record on the hit path (V = 1: the minimizer's preferred place in its bucket asked for first, the unitig's packed bounds
# requested beside the pool words) against the round-5 look-ups (SQ_SEED_V=0): the seeding tests, then the bench both ways
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r6z; mkdir -p $O; cd $R
python -c "import torch" > /dev/null 2>&1
timeout -k 5 900 python -m pytest tests/test_map_gpu.py tests/test_index.py tests/test_exhaustive.py tests/test_long_reads.py tests/test_c1.py tests/test_scale_gpu.py -m gpu -x -q > $O/gputests.txt 2>&1; grep -E "passed|failed|error" $O/gputests.txt | tail -3
run() {  # label, env...
  local lab=$1; shift
  env "$@" timeout -k 5 400 python bench.py --steps 10 --warmup 2 --no-extras --cpu-sample 0 --fastq-pairs 0 --index-cache /tmp/ixc > $O/b_$lab.json 2> $O/b_$lab.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/b_$lab.json").read().strip().splitlines()[-1])
    print("$lab:", d["value"], "map_eq_s", d["breakdown"]["map_eq_s"], "tail", d["breakdown"]["tail_s(eq_export+merge+normalize+EM)"], {k: v["avg_ms"] for k, v in d["stages"].items() if k in ("k_seed", "k_mems", "k_score", "eq_static", "eq_table", "eq_flags_scan")}, "mini", d["stages"]["eq_mini_batches"]["ms_total"], "fills", d["breakdown"]["stats"].get("filter_fills"))
except Exception as e:
    print("$lab: failed", e); print(open("$O/b_$lab.err").read()[-600:])
PY
}
run v1 SQ_SEED_V=1
run v0 SQ_SEED_V=0
run v1b SQ_SEED_V=1
run v0b SQ_SEED_V=0
echo done
