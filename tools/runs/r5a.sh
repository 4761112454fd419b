#!/bin/bash
# round 5, call A: k_seed2 variants (SQ_SEED_V: 0 = round 4's k_seed, 1 = read words in LDS, 2 = + filter block in LDS, 3 = + minimizer table,
# 4 = read words + minimizer table, 5 = 3 with the block moved by LDS-DMA, 6 = 2 with LDS-DMA; SQ_SEED_LW = 8: eight-word LDS column) — parity first, then time.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r5a; mkdir -p $O
cd $R
for v in 3 5 2; do
  SQ_SEED_V=$v timeout 600 python -m pytest tests/test_map_gpu.py tests/test_exhaustive.py tests/test_long_reads.py -m gpu -x -q > $O/pytest_v$v.log 2>&1
  echo "V=$v: $(tail -1 $O/pytest_v$v.log)"
done
Q="--no-extras --cpu-sample 0 --fastq-pairs 0 --index-cache /tmp/ixc"
timeout 300 python bench.py --steps 2 --warmup 1 $Q > $O/b_warm.json 2> $O/b_warm.err
for v in 0 1 2 3 4 5 6; do
  SQ_SEED_V=$v timeout 300 python bench.py --steps 6 --warmup 2 $Q > $O/b_v$v.json 2> $O/b_v$v.err
done
for v in 3 5; do
  SQ_SEED_V=$v SQ_SEED_LW=8 timeout 300 python bench.py --steps 6 --warmup 2 $Q > $O/b_v${v}_lw8.json 2> $O/b_v${v}_lw8.err
  SQ_SEED_V=$v SQ_SEED_BPC=8 timeout 300 python bench.py --steps 6 --warmup 2 $Q > $O/b_v${v}_bpc8.json 2> $O/b_v${v}_bpc8.err
done
python - <<'PY'
import json, glob, os
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.getcwd()), "gpurun_out", "r5a")
for f in sorted(glob.glob(O + "/b_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        st = d["stages"]
        print("%-16s value %7.1f  k_seed %6.3f ms  k_mems %6.3f  map_eq_s %.4f  em_call_s %.4f" % (os.path.basename(f), d["value"], st["k_seed"]["avg_ms"], st["k_mems"]["avg_ms"], d["breakdown"]["map_eq_s"], d["breakdown"]["em_call_s"]))
    except Exception as e:
        print(os.path.basename(f), "unreadable:", e)
PY
echo done
