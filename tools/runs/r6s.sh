#!/bin/bash
# round 6, call S: the CLI on the small gzip fixture again and again — does it ever hang, and where
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r6s; mkdir -p $O; cd $R
which gdb eu-stack 2>&1 | head -2
G=tests/golden; E=salmon_amd/bin/salmon-hip
$E index -t $G/transcripts.fa.gz -i /tmp/idxs -p 2 > /dev/null 2>&1
hang=0
for i in $(seq 1 200); do
  SQ_EXIT_TRACE=1 $E quant -i /tmp/idxs -l IU -1 $G/reads_1.fq.gz -2 $G/reads_2.fq.gz -o /tmp/outs --dumpEqWeights --numBootstraps 3 --seed 7 > $O/run.log 2>&1 &
  pid=$!
  for t in $(seq 1 40); do sleep 0.5; kill -0 $pid 2>/dev/null || break; done
  if kill -0 $pid 2>/dev/null; then
    hang=$((hang+1)); echo "run $i hangs (pid $pid)" | tee -a $O/hangs.txt; tail -4 $O/run.log; for t in /proc/$pid/task/*; do echo "== $(cat $t/comm) $(cat $t/wchan 2>/dev/null)"; cat $t/stack 2>/dev/null | head -8; done >> $O/hangs.txt 2>&1
    for t in /proc/$pid/task/*; do echo "$(cat $t/comm) $(cat $t/wchan 2>/dev/null) state $(grep State $t/status | cut -f2)"; done | sort | uniq -c | head -30
    if which gdb > /dev/null 2>&1; then timeout 60 gdb -p $pid -batch -ex "thread apply all bt 12" 2>/dev/null | grep -E "^Thread|^#" | head -150 > $O/bt_$i.txt; head -120 $O/bt_$i.txt; fi
    kill -9 $pid; [ $hang -ge 2 ] && break
  fi
done
echo "hangs: $hang of $i runs" | tee $O/summary.txt
echo done
