#!/bin/bash
# the channel churn of the GPU suite failed in session 50's suite run: which expectation?  Ten runs, the FAIL lines and the churn summary of each
cd /root/repo; mkdir -p gpurun_out/r06
{
for i in 1 2 3 4 5 6 7 8 9 10; do
  t0=$(date +%s)
  ( cd /tmp && /root/repo/tests/host/test_channel churn 32 8 2.4 1 > /tmp/churn_$i.out 2> /tmp/churn_$i.err; echo "run $i rc $? ($(( $(date +%s) - t0 )) s)" )
  grep -E "^FAIL|churn:|CHANNEL" /tmp/churn_$i.out | cut -c1-1500
done
} > gpurun_out/r06/session51.txt 2>&1
grep -E "rc |FAIL" gpurun_out/r06/session51.txt | cut -c1-600
