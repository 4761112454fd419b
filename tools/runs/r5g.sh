#!/bin/bash
# round 5, call G: the CIGAR error model on the device against the checker; the whole GPU suite; what the box gives a process (CPU quota, NUMA)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r5g; mkdir -p $O; cd $R
{ echo "nproc: $(nproc)"; echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"; echo "cfs_quota: $(cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>/dev/null) / $(cat /sys/fs/cgroup/cpu/cpu.cfs_period_us 2>/dev/null)";
  echo "cpuset: $(cat /sys/fs/cgroup/cpuset.cpus.effective 2>/dev/null)"; echo "affinity: $(taskset -p $$ 2>/dev/null)"; lscpu | grep -E "Model name|Socket|Core|Thread|NUMA" ; free -g | head -2; } > $O/box.txt 2>&1
cat $O/box.txt
python -c "import torch" > /dev/null 2>&1
timeout 600 python -m pytest tests/test_error_model.py tests/test_alignment_mode.py -m gpu -x -q > $O/pytest_err.log 2>&1; tail -3 $O/pytest_err.log
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu_all.log 2>&1; tail -3 $O/pytest_gpu_all.log
# how many cores does the inflating pool really get?  (one busy loop per thread for a second)
python - <<'PY'
import threading, time, os
def spin(out, i):
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < 1.0: n += 1
    out[i] = n
import multiprocessing as mp
def proc(q):
    t0 = time.process_time(); w0 = time.perf_counter(); x = 0
    while time.perf_counter() - w0 < 1.0: x += 1
    q.put(time.process_time() - t0)
for k in (16, 64, 128):
    q = mp.Queue(); ps = [mp.Process(target=proc, args=(q,)) for _ in range(k)]
    [p.start() for p in ps]; cpu = sum(q.get() for _ in ps); [p.join() for p in ps]
    print("%d busy processes for 1 s of wall time got %.1f CPU-seconds" % (k, cpu))
PY
echo done
