#!/bin/bash
# round 4, call D: keep-warm wave on/off (tail timings), then configs[3] at FULL size: 3.1 Gnt decoy genome, 20 M 2x150 pairs, parity sample
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
cd $R; mkdir -p gpurun_out/r4d; O=$R/gpurun_out/r4d
X="--steps 20 --warmup 1 --no-extras --cpu-sample 0 --fastq-pairs 0 --index-cache /tmp/ixc"
SQ_TIMING=1 timeout 300 python bench.py $X > $O/b_warm1.json 2> $O/b_warm1.err
SQ_TIMING=1 SQ_KEEP_WARM=0 timeout 300 python bench.py $X > $O/b_warm0.json 2> $O/b_warm0.err
X10="--steps 2 --warmup 1 --no-extras --cpu-sample 0 --fastq-pairs 0 --index-cache /tmp/ixc"
timeout 300 python bench.py $X10 > $O/b10_warm1.json 2> $O/b10_warm1.err
SQ_KEEP_WARM=0 timeout 300 python bench.py $X10 > $O/b10_warm0.json 2> $O/b10_warm0.err
( while true; do free -g | sed -n 2p; sleep 20; done ) > $O/mem_trace.txt 2>&1 &
MT=$!
timeout 1700 python bench.py --workload c4 --genome-gnt 3.1 --steps 5 --warmup 1 --cpu-sample 200000 > $O/bench_c4_full.json 2> $O/bench_c4_full.err
kill $MT
echo done
