#!/bin/bash
# round 6, call V: the index builder's k-mer table on the device (tests, then configs[3] at full size with the builder's phases timed), what the box gives a container
# (memory, /dev/shm), and configs[2] (N = 1) / configs[4] on the round's kernels
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r6v; mkdir -p $O; cd $R
python -c "import torch" > /dev/null 2>&1
timeout -k 5 600 python -m pytest tests/test_index_device_build.py tests/test_index.py -x -q 2>&1 | tail -3
{ echo "memory.max: $(cat /sys/fs/cgroup/memory.max 2>/dev/null)"; echo "memory.current: $(cat /sys/fs/cgroup/memory.current 2>/dev/null)"; df -h /dev/shm /tmp | cat; free -g | head -2; } > $O/box.txt 2>&1; cat $O/box.txt
C4="--workload c4 --genome-gnt 3.1 --warmup 1 --no-extras --fastq-pairs 0 --index-cache /tmp/ixc4"
SQ_TIMING=1 timeout 1500 python bench.py $C4 --steps 5 --cpu-sample 200000 > $O/bench_c4_full.json 2> $O/bench_c4_full.err; grep "sq-timing\] index" $O/bench_c4_full.err | tee $O/index_phases_c4.txt; tail -c 300 $O/bench_c4_full.err
timeout 400 python bench.py --workload c3 --no-extras --cpu-sample 0 --fastq-pairs 0 --index-cache /tmp/ixc > $O/bench_c3_n1.json 2> $O/bench_c3_n1.err
timeout 400 python bench.py --workload c5 --no-extras --cpu-sample 200000 --fastq-pairs 0 --index-cache /tmp/ixc > $O/bench_c5.json 2> $O/bench_c5.err
cd /tmp
timeout -k 5 900 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python $R/bench.py $C4 --steps 3 --cpu-sample 0 > $O/kt_c4.json 2> $O/kt_c4.err
db=$(find $O/kt -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/kstats.py $db "" 70 > $O/kernel_stats_c4_full.txt; rm -rf $O/kt
head -24 $O/kernel_stats_c4_full.txt | cut -c1-170
python - <<PY
import json
for n in ("c3_n1", "c5", "c4_full"):
    try:
        d = json.loads(open("$O/bench_%s.json" % n).read().strip().splitlines()[-1])
        print(n, d["value"], d["ms_per_step"], d["breakdown"]["map_eq_s"], d["breakdown"]["em_iters"], (d.get("parity_check") or {}).get("equal"), {k: d["roofline"].get(k) for k in ("kernel", "frac", "avg_launch_ms")}, d["breakdown"]["index_build_s"])
        print({k: v["avg_ms"] for k, v in d["stages"].items()})
    except Exception as e: print(n, "failed", e)
PY
echo done
