#!/bin/bash
# round 6, call ZC: the minimizer table with the unitig's packed bounds in the entry (mtab2: a hit costs two sectors — table, string pool — instead of three) against
# the round-5 table + bounds array (SQ_SEED_MTAB2=0): seeding tests, then the bench both ways
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r6zc; mkdir -p $O; cd $R
python -c "import torch" > /dev/null 2>&1
timeout -k 5 900 python -m pytest tests/test_map_gpu.py tests/test_index.py tests/test_exhaustive.py tests/test_long_reads.py tests/test_c1.py tests/test_poison_gpu.py tests/test_scale_gpu.py -m gpu -x -q > $O/gputests.txt 2>&1; grep -E "passed|failed|error" $O/gputests.txt | tail -1
run() {  # label, env...
  local lab=$1; shift
  env "$@" timeout -k 5 400 python bench.py --steps 10 --warmup 2 --no-extras --cpu-sample 0 --fastq-pairs 0 --index-cache /tmp/ixc > $O/b_$lab.json 2> $O/b_$lab.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/b_$lab.json").read().strip().splitlines()[-1])
    print("$lab:", d["value"], "map_eq_s", d["breakdown"]["map_eq_s"], {k: v["avg_ms"] for k, v in d["stages"].items() if k in ("k_pack", "k_seed", "k_mems", "k_score", "eq_static")}, "mini", d["stages"]["eq_mini_batches"]["ms_total"], "hbm", d["config"]["index_hbm_bytes"])
except Exception as e:
    print("$lab: failed", e); print(open("$O/b_$lab.err").read()[-600:])
PY
}
run mtab2 SQ_X=1
run mtab SQ_SEED_MTAB2=0
run mtab2b SQ_X=1
run mtabb SQ_SEED_MTAB2=0
echo done
