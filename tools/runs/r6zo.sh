#!/bin/bash
# round 6, call ZO: k_pack with a wave's text staged through LDS (k_pack8_staged) against the thread-per-end kernel: reader / mapping / long-read tests, c2 stage timers both ways
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r6zo; mkdir -p $O; cd $R
python -c "import torch" > /dev/null 2>&1
timeout -k 5 1200 python -m pytest tests/test_map_gpu.py tests/test_long_reads.py tests/test_reader_gpu.py tests/test_properties_gpu.py tests/test_c1.py -m gpu -x -q > $O/gputests.txt 2>&1; grep -E "passed|failed|error" $O/gputests.txt | tail -3
run() { local lab=$1; shift
  env "$@" timeout -k 5 400 python bench.py --steps 10 --warmup 2 --no-extras --cpu-sample 0 --fastq-pairs 0 --index-cache /tmp/ixc > $O/b_$lab.json 2> $O/b_$lab.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/b_$lab.json").read().strip().splitlines()[-1])
    print("$lab:", d["value"], "map_eq_s", d["breakdown"]["map_eq_s"], {k: v["avg_ms"] for k, v in d["stages"].items() if k in ("k_pack", "k_seed", "k_mems", "k_select", "k_score")})
except Exception as e:
    print("$lab failed", e); print(open("$O/b_$lab.err").read()[-600:])
PY
}
run staged SQ_X=1
run direct SQ_PACK_DIRECT=1
run staged2 SQ_X=1
run direct2 SQ_PACK_DIRECT=1
echo done
