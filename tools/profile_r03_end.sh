#!/bin/bash
# Round-3 closing measurements.  part A: the whole GPU test suite + the c2 bench lines; part B: the other workloads, kernel summaries, counter passes.
part=${1:-A}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r3end; mkdir -p $O
cd $R
if [ "$part" = A ]; then
  timeout 1200 python -m pytest tests -m gpu -q --timeout 300 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
  timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_c2_steps20.json 2> $O/bench_c2_steps20.err
  timeout 600 python bench.py > $O/bench_c2_default.json 2> $O/bench_c2_default.err
else
  timeout 600 python bench.py --workload c5 > $O/bench_c5.json 2> $O/bench_c5.err
  timeout 900 python bench.py --workload c4 --steps 6 --warmup 1 > $O/bench_c4.json 2> $O/bench_c4.err
  timeout 600 python bench.py --workload c2s --steps 10 --warmup 1 --fastq-pairs 0 > $O/bench_c2s.json 2> $O/bench_c2s.err
  cd /tmp
  for w in c2 c5 c4; do st=3; [ $w = c5 ] && st=4
    timeout -k 5 600 rocprofv3 --kernel-trace --stats -d $O/kt_$w -o kt -- python $R/bench.py --workload $w --steps $st --warmup 1 --cpu-sample 0 --fastq-pairs 0 > $O/kt_$w.json 2> $O/kt_$w.err
    db=$(find $O/kt_$w -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/kstats.py $db "" 45 > $O/kernel_stats_$w.txt; rm -rf $O/kt_$w
  done
  run() { name=$1; shift
    timeout -k 5 240 rocprofv3 --kernel-trace --pmc "$@" -d $O/p_$name -o pmc --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --cpu-sample 0 --fastq-pairs 0 > /dev/null 2> $O/p_$name.err
    python $R/tools/pmc_summary.py $O/p_$name 40 $O/pmc_$name.json > $O/pmc_$name.txt; rm -rf $O/p_$name; }
  run FETCH_SIZE FETCH_SIZE
  run WRITE_SIZE WRITE_SIZE
  run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
  run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA
fi
