#!/bin/bash
# round 4, call N: the final measurements.  Kernel trace and the two PMC passes of c2 first (their summaries go to profiles/ on the box, so that the
# bench line that follows carries the traffic figure of exactly these kernel sources), then the bench lines of c2 (defaults, and the driver's command),
# c2s, c5 and c3 on one GPU, then the whole GPU test suite.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r4n; mkdir -p $O
Q="--no-extras --cpu-sample 0 --fastq-pairs 0 --index-cache /tmp/ixc"
cd $R; timeout 300 python bench.py --steps 2 --warmup 1 $Q > $O/b_warm.json 2> $O/b_warm.err     # builds the index cache; a box's first run is slower
cd /tmp
timeout -k 5 400 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python $R/bench.py --steps 6 --warmup 1 $Q > $O/kt_c2.json 2> $O/kt_c2.err
db=$(find $O/kt -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/kstats.py $db "" 70 > $O/r04_kernel_stats_c2_final.txt; rm -rf $O/kt
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 5 300 rocprofv3 --kernel-trace --pmc $c -d $O/pmc_$c -o pmc --output-format csv -- python $R/bench.py --steps 2 --warmup 1 $Q > /dev/null 2> $O/pmc_$c.err
  python $R/tools/pmc_summary.py $O/pmc_$c 60 $O/pmc_$c.json > $O/pmc_$c.txt; rm -rf $O/pmc_$c
done
cd $R
python tools/pmc_traffic.py $O/pmc_FETCH_SIZE.json $O/pmc_WRITE_SIZE.json $O/r04_pmc_traffic.json "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE, separate passes, bench.py --steps 2 --warmup 1 (c2, 5 000 000 pairs per launch), tools/runs/r4n.sh" 5000000 > $O/pmc_traffic.log 2>&1
cp $O/r04_pmc_traffic.json $O/r04_kernel_stats_c2_final.txt $R/profiles/ 2>/dev/null
timeout 900 python bench.py --index-cache /tmp/ixc > $O/bench_c2_default.json 2> $O/bench_c2_default.err
timeout 400 python bench.py --steps 20 --warmup 5 $Q > $O/bench_c2_driver_cmd.json 2> $O/bench_c2_driver_cmd.err
timeout 400 python bench.py --workload c2s --no-extras --cpu-sample 500000 --fastq-pairs 0 > $O/bench_c2s.json 2> $O/bench_c2s.err
timeout 400 python bench.py --workload c5 --no-extras --cpu-sample 500000 --fastq-pairs 0 > $O/bench_c5.json 2> $O/bench_c5.err
timeout 400 python bench.py --workload c3 $Q > $O/bench_c3_n1.json 2> $O/bench_c3_n1.err
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1
tail -3 $O/pytest_gpu.log
echo done
