#!/bin/bash
# last check of the round at HEAD: the whole GPU suite and the driver's bench command
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r3z; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q --timeout 300 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_c2_steps20.json 2> $O/bench_c2_steps20.err
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1
