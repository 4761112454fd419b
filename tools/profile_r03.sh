#!/bin/bash
# Round-3 counter passes on the GPU box (kernel-trace + pmc only, one rocprofv3 run per counter set), workload = bench.py default (c2) at --steps 2.
#   tools/profile_r03.sh <tag>  -> gpurun_out/<tag>/pmc_{FETCH_SIZE,WRITE_SIZE,sq1,sq2}.{json,txt} + gather calibration
tag=${1:-r3pmc}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp
run() {   # name, counters...
  name=$1; shift
  timeout -k 5 240 rocprofv3 --kernel-trace --pmc "$@" -d $O/p_$name -o pmc --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --cpu-sample 0 --fastq-pairs 0 > /dev/null 2> $O/p_$name.err
  python $R/tools/pmc_summary.py $O/p_$name 40 $O/pmc_$name.json > $O/pmc_$name.txt
  rm -rf $O/p_$name
}
run FETCH_SIZE FETCH_SIZE
run WRITE_SIZE WRITE_SIZE
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA
# calibration of FETCH_SIZE on random 8-byte loads of 64-byte lines (known byte count): tools/gather_bench
python -c "import sys; sys.path.insert(0, '$R'); from salmon_amd import build; build.build_microbench()"
timeout -k 5 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/p_gather -o pmc --output-format csv -- $R/tools/_build/gather_bench > $O/gather_bench.txt 2> $O/p_gather.err
python $R/tools/pmc_summary.py $O/p_gather 10 $O/pmc_gather.json > $O/pmc_gather.txt
rm -rf $O/p_gather
