#!/bin/bash
# round 6, call ZA: the seed kernel's variants — 0: round 5's look-ups; 1: preferred place of the minimizer table + packed bounds beside the pool words (call Z: slower, a
# wave nearly always holds a lane whose place is taken and waits for its second trip); 3: whole bucket + packed bounds; 4: k_seed3 (filter blocks of sixteen lanes per
# instruction) + whole bucket + packed bounds; 2: k_seed3 + preferred place
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r6za; mkdir -p $O; cd $R
python -c "import torch" > /dev/null 2>&1
for v in 4 2 3; do SQ_SEED_V=$v timeout -k 5 600 python -m pytest tests/test_map_gpu.py tests/test_exhaustive.py tests/test_long_reads.py tests/test_c1.py -m gpu -x -q > $O/gputests_v$v.txt 2>&1; echo "V=$v: $(grep -E 'passed|failed|error' $O/gputests_v$v.txt | tail -1)"; done
run() {  # label, env...
  local lab=$1; shift
  env "$@" timeout -k 5 400 python bench.py --steps 10 --warmup 2 --no-extras --cpu-sample 0 --fastq-pairs 0 --index-cache /tmp/ixc > $O/b_$lab.json 2> $O/b_$lab.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/b_$lab.json").read().strip().splitlines()[-1])
    print("$lab:", d["value"], "map_eq_s", d["breakdown"]["map_eq_s"], {k: v["avg_ms"] for k, v in d["stages"].items() if k in ("k_seed", "k_mems", "k_score")}, "fills", d["breakdown"]["stats"].get("filter_fills"))
except Exception as e:
    print("$lab: failed", e); print(open("$O/b_$lab.err").read()[-600:])
PY
}
run v0 SQ_SEED_V=0
run v3 SQ_SEED_V=3
run v4 SQ_SEED_V=4
run v2 SQ_SEED_V=2
run v0b SQ_SEED_V=0
run v3b SQ_SEED_V=3
run v4b SQ_SEED_V=4
echo done
