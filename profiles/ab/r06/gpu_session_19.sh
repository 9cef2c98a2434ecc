#!/bin/bash
# closed loop, config 2: work-groups of 640 / 768 / 896 threads (10 / 12 / 14 waves: the per-wave fixed cost paid fewer times, three waves per SIMD still covering latencies?)
cd /root/repo; mkdir -p gpurun_out/r06
{
for tag in shipped t896pf1 t768pf1 t768pf2 t640pf2 shipped t768pf1; do
  if [ $tag = shipped ]; then unset GSH_LIB_PATH; else export GSH_LIB_PATH=/root/repo/build/variants/lib_$tag.so; fi
  echo "== $tag"; timeout 300 python profiles/ab/r06/closed_loop_split.py 2>&1 | grep -v amdgpu | grep "1 work-group\|2 work-group" | cut -c1-200
done
export GSH_LIB_PATH=/root/repo/build/variants/lib_t768pf1.so
timeout 600 python -m pytest tests/test_tracking_loop_gpu.py tests/test_tracking_live_gpu.py -x -q -m gpu 2>&1 | tail -3
} > gpurun_out/r06/session19.txt 2>&1
cat gpurun_out/r06/session19.txt
