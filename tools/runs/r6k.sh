#!/bin/bash
# round 6, call K: configs[3] at full size with the five-word k_seed2 (2 x 150), LW 5 against LW 8, + the 2x150 / 2x250 / 2x75 parity tests
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r6k; mkdir -p $O; cd $R
python -c "import torch" > /dev/null 2>&1
timeout -k 5 600 python -m pytest tests/test_map_gpu.py -m gpu -x -q -k "noisy_and_long or c4_shape" > $O/gputests.txt 2>&1; tail -3 $O/gputests.txt
C4="--workload c4 --genome-gnt 3.1 --warmup 1 --no-extras --fastq-pairs 0 --index-cache /tmp/ixc4 --steps 4 --cpu-sample 0"
for lw in 5 8; do
SQ_SEED_LW=$lw timeout 1500 python bench.py $C4 > $O/bench_c4_lw$lw.json 2> $O/bench_c4_lw$lw.err; tail -c 300 $O/bench_c4_lw$lw.err
python - <<PY
import json
d = json.loads(open("$O/bench_c4_lw$lw.json").read().strip().splitlines()[-1])
print("LW $lw", d["value"], d["ms_per_step"], {k: v["avg_ms"] for k, v in d["stages"].items() if k in ("k_seed", "k_mems", "large_ends", "k_pack", "k_score", "k_dp")})
PY
done
echo done
