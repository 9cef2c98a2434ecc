#!/bin/bash
# batched correlator: the paired trip's look-ups issued ahead of the wait for the samples and the rotations (GSH_MC_EARLY_CODES), beside the shipped order
cd /root/repo; mkdir -p gpurun_out/r06
{
for tag in shipped ec shipped ec shipped ec; do
  if [ $tag = shipped ]; then unset GSH_LIB_PATH; else export GSH_LIB_PATH=/root/repo/build/variants/lib_$tag.so; fi
  echo "== $tag"; timeout 300 python profiles/ab/r06/mcorr_n.py 25000 2>&1 | grep -v amdgpu | tail -1
done
export GSH_LIB_PATH=/root/repo/build/variants/lib_ec.so
timeout 600 python -m pytest tests/test_tracking_gpu.py -x -q -m gpu 2>&1 | tail -3
} > gpurun_out/r06/session20.txt 2>&1
cat gpurun_out/r06/session20.txt
