#!/bin/bash
# round 4, call A: fence-free grid barrier microbenchmark, the new bench line (100 M-pair job + extras), W / CU-partition experiments
cd /root/repo; mkdir -p gpurun_out/r4a; O=gpurun_out/r4a
timeout 900 python bench.py --steps 20 --warmup 2 --index-cache /tmp/ixc > $O/bench_base.json 2> $O/bench_base.err
X="--steps 20 --warmup 1 --no-extras --cpu-sample 0 --fastq-pairs 0 --index-cache /tmp/ixc"
timeout 300 python bench.py $X --inflight 32 > $O/b_w32.json 2> $O/b_w32.err
timeout 300 python bench.py $X --inflight 64 > $O/b_w64.json 2> $O/b_w64.err
SQ_MAP_ALL_CUS=1 timeout 300 python bench.py $X --inflight 32 > $O/b_w32_all.json 2> $O/b_w32_all.err
SQ_MAP_ALL_CUS=1 timeout 300 python bench.py $X --inflight 64 > $O/b_w64_all.json 2> $O/b_w64_all.err
SQ_MAP_ALL_CUS=1 timeout 300 python bench.py $X > $O/b_w8_all.json 2> $O/b_w8_all.err
SQ_EQ_CUS=32 timeout 300 python bench.py $X --inflight 32 > $O/b_w32_cu32.json 2> $O/b_w32_cu32.err
SQ_EQ_CUS=32 timeout 300 python bench.py $X --inflight 64 > $O/b_w64_cu32.json 2> $O/b_w64_cu32.err
SQ_EQ_CUS=16 timeout 300 python bench.py $X --inflight 64 > $O/b_w64_cu16.json 2> $O/b_w64_cu16.err
echo done
