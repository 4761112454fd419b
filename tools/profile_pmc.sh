#!/bin/bash
# Arbitrary counter passes on the GPU box (kernel-trace + pmc only), one rocprofv3 run per quoted counter set.
# Usage: tools/profile_pmc.sh tag batch "SET ONE" "SET TWO" ...  -> gpurun_out/<tag>_p<i>.{txt,json}
tag=$1; batch=$2; shift 2
export TMPDIR=/tmp
out=$PWD/gpurun_out; mkdir -p $out
cd /tmp
i=0
for set in "$@"; do
  i=$((i+1))
  timeout -k 5 120 rocprofv3 --kernel-trace --pmc $set -d $out/${tag}_p$i -o pmc --output-format csv -- python $out/../bench.py --batch $batch --steps 2 --warmup 1 --cpu-sample 0 --fastq-pairs 0 > /dev/null 2> $out/${tag}_p$i.err
  python $out/../tools/pmc_summary.py $out/${tag}_p$i 24 $out/${tag}_p$i.json > $out/${tag}_p$i.txt
  rm -rf $out/${tag}_p$i
done
