#!/bin/bash
# round 6, call ZR: the same 100 M-pair job cut into smaller sq_map_batch calls (the first call's mapping and the last call's eq stage overlap with nothing: ~13 ms each at 5 M pairs per call)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r6zr; mkdir -p $O; cd $R
python -c "import torch" > /dev/null 2>&1
run() { local lab=$1; shift
  timeout -k 5 500 python bench.py --steps 20 --warmup 5 --no-extras --cpu-sample 0 --fastq-pairs 0 --index-cache /tmp/ixc "$@" > $O/b_$lab.json 2> $O/b_$lab.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/b_$lab.json").read().strip().splitlines()[-1])
    print("$lab:", d["value"], "map_eq_s", d["breakdown"]["map_eq_s"], "tail", d["breakdown"]["tail_s(eq_export+merge+normalize+EM)"], "eqf", d["breakdown"]["eq_finish_s"], "em", d["breakdown"]["em_call_s"], d["breakdown"]["em_iters"], d["config"].get("pairs_per_step"), d["config"].get("job_pairs"))
except Exception as e:
    print("$lab failed", e); print(open("$O/b_$lab.err").read()[-600:])
PY
}
run b5m_s1 --batch 5000000 --sub 1
run b2p5m_s2 --batch 2500000 --sub 2
run b1p25m_s4 --batch 1250000 --sub 4
run b1m_s5 --batch 1000000 --sub 5
run b5m_s1_again --batch 5000000 --sub 1
run b2p5m_s2_again --batch 2500000 --sub 2
echo done
