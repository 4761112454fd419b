#!/bin/bash
# round 5, call M: the whole GPU suite (every failure listed), smoke(), and the driver's bench command with the committed PMC profile attached
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r5m; mkdir -p $O; cd $R
python -c "import torch" > /dev/null 2>&1
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu_all.log 2>&1; grep -a "passed\|failed" $O/pytest_gpu_all.log | tail -2; grep -a "^FAILED" $O/pytest_gpu_all.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; python - <<PY
import json; d = json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1]); r = d["roofline"]
print(d["value"], d["ms_per_step"], {k: r.get(k) for k in ("kernel", "frac", "traffic", "avg_launch_ms")}, r.get("chain_pair_note", "")[:120])
print({k: (v.get("value") if isinstance(v, dict) else v) for k, v in d["from_fastq"].items() if k in ("plain", "gzip", "bgzf", "compressed_error")})
PY
echo done
