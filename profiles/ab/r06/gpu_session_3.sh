#!/bin/bash
# round 6, session 3: chains of jobs per work-group
cd /root/repo; export TMPDIR=/tmp
mkdir -p gpurun_out/r06
{
python -m pytest tests/test_tracking_gpu.py tests/test_tracking_configs_gpu.py -m gpu -x -q 2>&1 | tail -5
python profiles/ab/r06/chain_ab.py
} > gpurun_out/r06/session3.txt 2>&1
grep -v amdgpu.ids gpurun_out/r06/session3.txt | tail -40
