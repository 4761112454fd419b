#!/bin/bash
# round 6, call ZE: configs[3] at full size — from how many MEMs on a read end takes the flat large-end passes (default: more than 1 024; k_mems<64,1024> is 7.4 ms per 4 M pairs,
# a lane per transcript group there): 1024 / 512 / 256 / 128 / 64; the index is built once and cached on the box
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r6ze; mkdir -p $O; cd $R
python -c "import torch" > /dev/null 2>&1
C4="--workload c4 --genome-gnt 3.1 --warmup 1 --no-extras --fastq-pairs 0 --index-cache /tmp/ixc4"
run() {  # label, extra bench args, env...
  local lab=$1; local extra=$2; shift; shift
  env "$@" timeout -k 5 1200 python bench.py $C4 --steps 5 $extra > $O/b_$lab.json 2> $O/b_$lab.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/b_$lab.json").read().strip().splitlines()[-1])
    print("$lab:", d["value"], "map_eq_s", d["breakdown"]["map_eq_s"], (d.get("parity_check") or {}).get("equal"), {k: v["avg_ms"] for k, v in d["stages"].items() if k in ("k_seed", "k_mems", "large_ends", "k_score", "k_dp")}, "idx", d["breakdown"]["index_build_s"], "chains", d["breakdown"]["stats"]["num_chains"], "alns", d["breakdown"]["stats"]["num_alignments"])
except Exception as e:
    print("$lab: failed", e); print(open("$O/b_$lab.err").read()[-600:])
PY
}
run t1024 "--cpu-sample 0" SQ_X=1
run t256 "--cpu-sample 200000" SQ_LG_THRESH=256
run t512 "--cpu-sample 0" SQ_LG_THRESH=512
run t128 "--cpu-sample 0" SQ_LG_THRESH=128
run t64 "--cpu-sample 0" SQ_LG_THRESH=64
run t1024b "--cpu-sample 0" SQ_X=1
echo done
