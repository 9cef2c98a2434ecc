#!/bin/bash
# (a) 200 000 points (S = 8) with all three of: non-temporal Z, ordered lanes, class-by-class sub-cells (build/variants/lib_nt8.so: -DGSH_OC_Z_NT_BELOW_S=9)
# (b) 128 000 points with the persistent sub-cell flavour once more (GSH_OC_DIT_PERSIST=1)
# (c) the channel churn under a loaded host (16 busy processes): twelve runs, every FAIL line kept
cd /root/repo; mkdir -p gpurun_out/r06
{
for rep in 1 2; do
echo "== shipped"; python profiles/ab/r06/acq_128k.py 200000:50e6 2>&1 | grep "^N ="
echo "== S = 8 with the hint, ordered, class by class"; GSH_LIB_PATH=/root/repo/build/variants/lib_nt8.so python profiles/ab/r06/acq_128k.py 200000:50e6 2>&1 | grep "^N ="
echo "== S = 8 with the hint, ordered, cell by cell"; GSH_OC_DIT_R_MAJOR=0 GSH_LIB_PATH=/root/repo/build/variants/lib_nt8.so python profiles/ab/r06/acq_128k.py 200000:50e6 2>&1 | grep "^N ="
done
echo "== 128 000 points, persistent sub-cells"; GSH_OC_DIT_PERSIST=1 python profiles/ab/r06/acq_128k.py 2>&1 | grep "^N ="
for j in $(seq 16); do ( python -c "
import time
t=time.time()
while time.time()-t<420: pass" & ) ; done
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do
  t0=$(date +%s)
  ( cd /tmp && /root/repo/tests/host/test_channel churn 32 8 2.4 1 > /tmp/churn_$i.out 2> /tmp/churn_$i.err; echo "loaded run $i rc $? ($(( $(date +%s) - t0 )) s)" )
  grep -E "^FAIL|dropped by the time limit" /tmp/churn_$i.out | cut -c1-2500
done
} > gpurun_out/r06/session53.txt 2>&1
cut -c1-1200 gpurun_out/r06/session53.txt
