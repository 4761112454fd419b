#!/bin/bash
# round 6, call P: k_seed2's persistent grid: 1536 blocks (256 CUs x 6) against what fits its 192-CU partition at six blocks per CU (1152) and other sizes
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r6p; mkdir -p $O; cd $R
python -c "import torch" > /dev/null 2>&1
for g in 6144 6656 7168 6656 6144; do
SQ_SEED_GRID=$g timeout -k 5 400 python bench.py --steps 10 --warmup 2 --no-extras --cpu-sample 0 --fastq-pairs 0 --index-cache /tmp/ixc > $O/b_$g.json 2> $O/b_$g.err
python - <<PY
import json
d = json.loads(open("$O/b_$g.json").read().strip().splitlines()[-1])
print("grid $g:", d["value"], d["breakdown"]["map_eq_s"], {k: v["avg_ms"] for k, v in d["stages"].items() if k in ("k_seed", "k_mems", "k_score", "eq_static")})
PY
done
echo done
