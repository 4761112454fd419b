#!/bin/bash
# Round profile on the GPU box: kernel trace + stats, then FETCH_SIZE and WRITE_SIZE in separate PMC passes (kernel-trace only).
# Usage: tools/profile_round.sh r02   -> gpurun_out/<tag>_*  (copy the summaries into profiles/)
tag=${1:-r02}
export TMPDIR=/tmp
out=$PWD/gpurun_out; mkdir -p $out
cd /tmp
timeout -k 5 200 rocprofv3 --kernel-trace --stats -d $out/${tag}_kt -o kt -- python $out/../bench.py --steps 10 --warmup 2 --cpu-sample 0 --fastq-pairs 0 > $out/${tag}_kt_bench.json 2> $out/${tag}_kt.err
db=$(find $out/${tag}_kt -name "*.db" | head -1)
[ -n "$db" ] && python $out/../tools/kstats.py $db "" 40 > $out/${tag}_kernel_stats.txt
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 5 150 rocprofv3 --kernel-trace --pmc $c -d $out/${tag}_pmc_$c -o pmc --output-format csv -- python $out/../bench.py --steps 2 --warmup 1 --cpu-sample 0 --fastq-pairs 0 > /dev/null 2> $out/${tag}_pmc_$c.err
  python $out/../tools/pmc_summary.py $out/${tag}_pmc_$c 60 $out/${tag}_pmc_$c.json > $out/${tag}_pmc_$c.txt
done
rm -rf $out/${tag}_kt $out/${tag}_pmc_FETCH_SIZE $out/${tag}_pmc_WRITE_SIZE
