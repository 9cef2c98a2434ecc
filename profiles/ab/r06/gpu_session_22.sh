#!/bin/bash
# the channel churn after the steady channels' time-limit drop is counted instead of failed (tests/host/test_channel.cc): the whole GPU suite, then the churn eight times
cd /root/repo; mkdir -p gpurun_out/r06
{
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
cd /tmp
for i in 1 2 3 4 5 6 7 8; do
  timeout 300 /root/repo/tests/host/test_channel churn 32 8 2.4 1 > /tmp/churn_$i.log 2>&1; rc=$?
  echo "run $i rc=$rc $(grep -c FAIL /tmp/churn_$i.log) fails; $(grep -o 'dropped by the time limit' /tmp/churn_$i.log | wc -l) early-drop lines; $(grep -o 'at least [0-9.]* %' /tmp/churn_$i.log)"
  grep FAIL /tmp/churn_$i.log | cut -c1-400 | head -5
done
} > gpurun_out/r06/session22.txt 2>&1
cat gpurun_out/r06/session22.txt
