#!/bin/bash
# Round-3 final measurements on the GPU box: the three workloads' bench lines, their rocprofv3 kernel-trace summaries, and the PMC passes for c2.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r3final; mkdir -p $O
cd $R
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_c2_driver_cmd.json 2> $O/bench_c2_driver_cmd.err
timeout 600 python bench.py > $O/bench_c2_default.json 2> $O/bench_c2_default.err
timeout 600 python bench.py --workload c5 > $O/bench_c5.json 2> $O/bench_c5.err
timeout 900 python bench.py --workload c4 --steps 6 --warmup 1 > $O/bench_c4.json 2> $O/bench_c4.err
timeout 600 python bench.py --workload c2s --steps 10 --warmup 1 --fastq-pairs 0 > $O/bench_c2s.json 2> $O/bench_c2s.err
cd /tmp
for w in c2 c5 c4; do st=3; [ $w = c5 ] && st=4
  timeout -k 5 600 rocprofv3 --kernel-trace --stats -d $O/kt_$w -o kt -- python $R/bench.py --workload $w --steps $st --warmup 1 --cpu-sample 0 --fastq-pairs 0 > $O/kt_$w.json 2> $O/kt_$w.err
  db=$(find $O/kt_$w -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/kstats.py $db "" 45 > $O/kernel_stats_$w.txt; rm -rf $O/kt_$w
done
