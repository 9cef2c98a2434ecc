#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r06
{
for m in 0 500 650; do
echo "== main share per mille: $m (0 = tuned table)"; GSH_TRK_SPLIT_MAIN=$m timeout 300 python profiles/ab/r06/closed_loop_split.py 2>&1 | grep -v amdgpu | cut -c1-75
done
} > gpurun_out/r06/session14.txt 2>&1
tail -20 gpurun_out/r06/session14.txt
