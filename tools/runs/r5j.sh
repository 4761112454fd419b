#!/bin/bash
# round 5, call J: the device inflater after input staging and deferred match stores — the kernel against zlib, its time, the reader tests, the from_fastq legs
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r5j; mkdir -p $O; cd $R
python -c "import torch" > /dev/null 2>&1
timeout 300 python -m pytest tests/test_inflate.py -m gpu -x -q > $O/pytest_inflate.log 2>&1; tail -3 $O/pytest_inflate.log
for reps in 8 16 40; do
cd /tmp && INFLATE_REPS=$reps timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_inflate -- python $R/tools/inflate_bench.py > $O/inflate_bench_$reps.log 2>&1; cd $R; grep -a const_q $O/inflate_bench_$reps.log | cut -c1-60
db=$(find $O/prof_inflate -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/kstats.py $db "" 1 | tail -1 | tee -a $O/inflate_kernel_stats.txt; rm -rf $O/prof_inflate
done
timeout 900 python -m pytest tests/test_reader_gpu.py -m gpu -x -q > $O/pytest_reader.log 2>&1; tail -3 $O/pytest_reader.log
if [ "$1" != "nobench" ]; then
SQ_READER_STATS=1 timeout 900 python bench.py --steps 4 --warmup 1 --no-extras --cpu-sample 0 --index-cache /tmp/ixc > $O/bench_fastq.json 2> $O/bench_fastq.err; python - <<PY
import json; d = json.loads(open("$O/bench_fastq.json").read().strip().splitlines()[-1]); print({k: (v.get("value") if isinstance(v, dict) else v) for k, v in d["from_fastq"].items() if k in ("plain", "gzip", "bgzf", "compressed_error")})
PY
grep -a "sq_dev_reader" $O/bench_fastq.err | tail -3
fi
echo done
