#!/bin/bash
# round 5, call D: compressed input through the device reader (tests + the bench's from-FASTQ legs on 20 M pairs), and which stream the optimiser should run on
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r5d; mkdir -p $O; cd $R
python -c "import torch" > /dev/null 2>&1
timeout 900 python -m pytest tests/test_reader_gpu.py tests/test_em.py tests/test_map_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
Q="--no-extras --cpu-sample 0 --fastq-pairs 0 --index-cache /tmp/ixc"
timeout 300 python bench.py --steps 2 --warmup 1 $Q > $O/warm.json 2> $O/warm.err
for rep in 1 2 3; do for v in 0 1; do
  SQ_EM_STREAM=$v SQ_TIMING=1 timeout 300 python bench.py --steps 8 --warmup 2 $Q > $O/em_s${v}_r$rep.json 2> $O/em_s${v}_r$rep.err
done; done
python - <<'PY'
import json, glob, os
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.getcwd()), "gpurun_out", "r5d")
for f in sorted(glob.glob(O + "/em_s*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); b = d["breakdown"]
        print("%-16s value %7.1f  em_call %.4f  em_device %.4f  set-up %.4f  eq_finish %.4f" % (os.path.basename(f), d["value"], b["em_call_s"], b["em_device_ms"] / 1e3, b["em_call_s"] - b["em_device_ms"] / 1e3, b["eq_finish_s"]))
    except Exception as e: print(os.path.basename(f), "unreadable", e)
PY
SQ_READER_STATS=1 timeout 900 python bench.py --steps 4 --warmup 1 --no-extras --cpu-sample 0 --index-cache /tmp/ixc > $O/fastq.json 2> $O/fastq.err
python - <<'PY'
import json, os
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.getcwd()), "gpurun_out", "r5d")
try:
    d = json.loads(open(O + "/fastq.json").read().strip().splitlines()[-1]); print(json.dumps(d.get("from_fastq"))[:1500])
except Exception as e: print("fastq leg unreadable", e)
PY
grep "sq_dev_reader" $O/fastq.err | tail -8
echo done
