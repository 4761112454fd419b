#!/bin/bash
# round 6, call L: the device gzip decoder against zlib (tests/test_gzip_dev.py)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r6l; mkdir -p $O; cd $R
python -c "import torch" > /dev/null 2>&1
timeout -k 5 600 python -m pytest tests/test_gzip_dev.py tests/test_inflate.py -x -q 2>&1 | tail -30 | cut -c1-400
echo done
