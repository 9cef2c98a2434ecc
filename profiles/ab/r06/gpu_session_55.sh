#!/bin/bash
# 50 000 points (S = 2, decimation in frequency, persistent sub-cells reading 800 KB each): does the working set of an XCD's work-groups in flight overrun its L2?
# work-groups per XCD x prefetch depth; and the decimation-in-time form from S = 2 with hint + ordered lanes + class-by-class walk
cd /root/repo; mkdir -p gpurun_out/r06
{
for w in 28 24 20 16 32; do for pf in 2 1 0; do
  echo "== work-groups per XCD $w, prefetch $pf"; GSH_OC_WG_PER_XCD=$w GSH_OC_PREFETCH=$pf python profiles/ab/r06/acq_128k.py 50000:50e6 2>&1 | grep "^N ="
done; done
echo "== decimation in time from S = 2"; GSH_OC_DIT_MIN_S=2 python profiles/ab/r06/acq_128k.py 50000:50e6 2>&1 | grep "^N ="
} > gpurun_out/r06/session55.txt 2>&1
cat gpurun_out/r06/session55.txt
