#!/bin/bash
# round 6, call ZP: k_select's look-back by a wave (64 descriptors per step) instead of one thread: the whole GPU suite, c2 stage timers (k_select was 1.02 - 1.11 ms in every earlier run)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r6zp; mkdir -p $O; cd $R
python -c "import torch" > /dev/null 2>&1
timeout -k 5 1500 python -m pytest tests -m gpu -x -q > $O/gputests.txt 2>&1; grep -E "passed|failed|error" $O/gputests.txt | tail -3
for i in 1 2; do
timeout -k 5 400 python bench.py --steps 10 --warmup 2 --no-extras --cpu-sample 0 --fastq-pairs 0 --index-cache /tmp/ixc > $O/b_$i.json 2> $O/b_$i.err
python - <<PY
import json
try:
    d = json.loads(open("$O/b_$i.json").read().strip().splitlines()[-1])
    print("c2 run $i:", d["value"], "map_eq_s", d["breakdown"]["map_eq_s"], {k: v["avg_ms"] for k, v in d["stages"].items()})
except Exception as e:
    print("failed", e); print(open("$O/b_$i.err").read()[-600:])
PY
done
echo done
