#!/bin/bash
# run a command; if it is still running after WAIT seconds, say where its threads wait (and a gdb backtrace if there is a gdb)
# usage: tools/hang_probe.sh OUTDIR WAIT command...
out=gpurun_out/$1; mkdir -p $out; wait_s=$2; shift 2
export PYTHONPATH=$PWD
"$@" > $out/probe.log 2>&1 &
pid=$!
for i in $(seq 1 $wait_s); do sleep 1; kill -0 $pid 2>/dev/null || break; done
if kill -0 $pid 2>/dev/null; then
  echo "STILL RUNNING after $wait_s s: threads and where they wait" >> $out/probe.log
  for p in $pid $(pgrep -P $pid); do
    for t in /proc/$p/task/*; do echo "pid $p tid $(basename $t) $(cat $t/comm) wchan=$(cat $t/wchan 2>/dev/null) $(grep State $t/status)"; done
    echo "--- maps (so files)"; grep -o "/[^ ]*\.so[^ ]*" /proc/$p/maps | sort -u | grep -v "python3.10/lib-dynload" | head -80
    which gdb > /dev/null 2>&1 && timeout 90 gdb -batch -ex "thread apply all bt 30" -p $p 2>&1 | grep -v "^\[New\|^warning" | head -200
  done >> $out/probe.log 2>&1
  kill -9 $pid $(pgrep -P $pid) 2>/dev/null
fi
tail -150 $out/probe.log
