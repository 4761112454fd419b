#!/bin/bash
# round 6, call ZU: bench.py's contract test and the smoke entry on the final tree
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r6zu; mkdir -p $O; cd $R
python -c "import torch" > /dev/null 2>&1
timeout -k 5 900 python -m pytest tests/test_bench_contract.py -m gpu -x -q > $O/gputests_bench.txt 2>&1; grep -E "passed|failed|error" $O/gputests_bench.txt | tail -3
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 python bench.py --steps 3 --warmup 1 --no-extras --cpu-sample 0 --fastq-pairs 0 --index-cache /tmp/ixc 2> $O/b.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], r['frac'], r['traffic'], r.get('traffic_over_alg_bytes'), r.get('valu_issue',{}).get('frac_of_issue_cycles'))"
echo done
