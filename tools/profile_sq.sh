#!/bin/bash
# SQ counter passes on the GPU box (kernel-trace + pmc only): wave cycles / wait buckets / instruction mix per kernel.
# Usage: tools/profile_sq.sh tag [batch]  -> gpurun_out/<tag>_sq{1,2}.{txt,json}
tag=${1:-sq}; batch=${2:-1000000}
export TMPDIR=/tmp
out=$PWD/gpurun_out; mkdir -p $out
cd /tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  timeout -k 5 120 rocprofv3 --kernel-trace --pmc $set -d $out/${tag}_sq$i -o pmc --output-format csv -- python $out/../bench.py --batch $batch --steps 2 --warmup 1 --cpu-sample 0 --fastq-pairs 0 > /dev/null 2> $out/${tag}_sq$i.err
  python $out/../tools/pmc_summary.py $out/${tag}_sq$i 24 $out/${tag}_sq$i.json > $out/${tag}_sq$i.txt
  rm -rf $out/${tag}_sq$i
done
