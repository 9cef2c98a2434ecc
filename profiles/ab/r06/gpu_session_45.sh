#!/bin/bash
# the four combinations (non-temporal Z or not) x (ordered or free-running lanes) at the other decimation-in-time shapes
cd /root/repo
mkdir -p gpurun_out/r06
{
for rep in 1 2; do
  echo "== NT, ordered"; python profiles/ab/r06/acq_128k.py 100000:25e6 200000:50e6 2>&1 | grep "^N ="
  echo "== NT, free lanes"; GSH_ACQ_DIT_ORDERED=0 python profiles/ab/r06/acq_128k.py 100000:25e6 200000:50e6 2>&1 | grep "^N ="
  echo "== no NT, ordered"; GSH_LIB_PATH=/root/repo/build/variants/lib_nont.so python profiles/ab/r06/acq_128k.py 100000:25e6 200000:50e6 2>&1 | grep "^N ="
  echo "== no NT, free lanes"; GSH_ACQ_DIT_ORDERED=0 GSH_LIB_PATH=/root/repo/build/variants/lib_nont.so python profiles/ab/r06/acq_128k.py 100000:25e6 200000:50e6 2>&1 | grep "^N ="
done
} > gpurun_out/r06/session45.txt 2>&1
cat gpurun_out/r06/session45.txt
