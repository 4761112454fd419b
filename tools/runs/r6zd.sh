#!/bin/bash
# round 6, call ZD: the k-mer filter's size against the 256 MB infinity cache (call ZC: a 32 MB array that answers from it beat a sector saved): 32 bits per k-mer (604 MB on c2,
# the default), 16 (302 MB), 8 (151 MB), 64; results are the same by construction (no false negatives): the bench's counters say so
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r6zd; mkdir -p $O; cd $R
python -c "import torch" > /dev/null 2>&1
run() {  # label, env...
  local lab=$1; shift
  env "$@" timeout -k 5 400 python bench.py --steps 10 --warmup 2 --no-extras --cpu-sample 0 --fastq-pairs 0 --index-cache /tmp/ixc > $O/b_$lab.json 2> $O/b_$lab.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/b_$lab.json").read().strip().splitlines()[-1]); st = d["breakdown"]["stats"]
    print("$lab:", d["value"], "map_eq_s", d["breakdown"]["map_eq_s"], {k: v["avg_ms"] for k, v in d["stages"].items() if k in ("k_seed", "k_mems", "k_score")}, "hbm", d["config"]["index_hbm_bytes"], "seeds", st["num_seeds"], "alns", st["num_alignments"], "fills", st.get("filter_fills"), "iters", d["breakdown"]["em_iters"])
except Exception as e:
    print("$lab: failed", e); print(open("$O/b_$lab.err").read()[-600:])
PY
}
run kf32 SQ_X=1
run kf16 SQ_KF_BITS=16
run kf8 SQ_KF_BITS=8
run kf64 SQ_KF_BITS=64
run kf32b SQ_X=1
run kf16b SQ_KF_BITS=16
run kf8b SQ_KF_BITS=8
echo done
