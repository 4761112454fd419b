#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r3f; mkdir -p $O
cd $R
python - > $O/reader_threads.txt 2>&1 <<'PY'
import os, sys, subprocess
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import reader_bench as rb
d = "/dev/shm/sq_reader_bench"; os.makedirs(d, exist_ok=True); N = 4000000
for gz in (False, True):
    ext = ".fq.gz" if gz else ".fq"
    rb.write(d + "/r_1" + ext, N, 1, gz); rb.write(d + "/r_2" + ext, N, 2, gz)
rb.bgzf_write(d + "/b_1.fq.gz", open(d + "/r_1.fq", "rb").read()); rb.bgzf_write(d + "/b_2.fq.gz", open(d + "/r_2.fq", "rb").read())
code = "import os,sys; sys.path.insert(0,os.getcwd()); sys.path.insert(0,'tools'); import reader_bench as rb; d='/dev/shm/sq_reader_bench'\nfor a,b in (('r_1.fq','r_2.fq'),('r_1.fq.gz','r_2.fq.gz'),('b_1.fq.gz','b_2.fq.gz')):\n    rb.drain(d+'/'+a,d+'/'+b,1000000); n,dt=rb.drain(d+'/'+a,d+'/'+b,1000000); print(a, round(n/dt/1e6,2), 'M pairs/s')"
for th in ("16", "32", "48", "64", "96"):
    env = dict(os.environ, SQ_READER_THREADS=th)
    print("SQ_READER_THREADS=" + th); sys.stdout.flush()
    print(subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True).stdout)
PY
