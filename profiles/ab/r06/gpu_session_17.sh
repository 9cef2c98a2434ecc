#!/bin/bash
# closed loop, config 2: 512-thread work-groups with one / two / three trips of loads in flight (256 VGPRs per wave at 8 waves per unit), beside the shipped 1 024;
# config 4 against the number of cooperating work-groups
cd /root/repo; mkdir -p gpurun_out/r06
{
for tag in shipped t512pf1 t512pf2 t512pf3 shipped t512pf2; do
  if [ $tag = shipped ]; then unset GSH_LIB_PATH; else export GSH_LIB_PATH=/root/repo/build/variants/lib_$tag.so; fi
  echo "== $tag"; timeout 300 python profiles/ab/r06/closed_loop_split.py 2>&1 | grep -v amdgpu | grep "1 work-group\|2 work-group" | cut -c1-110
done
unset GSH_LIB_PATH
echo "== config 4, cooperating work-groups"
timeout 600 python profiles/ab/r06/closed_loop_config4_split.py 2>&1 | grep -v amdgpu
} > gpurun_out/r06/session17.txt 2>&1
cat gpurun_out/r06/session17.txt
