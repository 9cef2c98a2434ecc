#!/bin/bash
# round 6, session 7: 128-thread work-groups against 256 at the shapes of configs 4 and 5 (five taps + fused data tap, windowed code tables, automatic splits)
cd /root/repo
mkdir -p gpurun_out/r06
{
for lib in "" build/variants/lib_t128.so; do
echo "== lib ${lib:-current (256 threads)}"
GSH_LIB_PATH=$lib python profiles/config_rates.py 2>&1 | grep -v amdgpu
done
} > gpurun_out/r06/session7.txt 2>&1
cat gpurun_out/r06/session7.txt | tail -40
