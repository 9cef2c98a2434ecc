#!/bin/bash
# round 6, session 2: carrier seeds from the work-group's factor table (GSH_MC_FAC=0 switches back): parity tests, then A/B of the launch time
cd /root/repo; export TMPDIR=/tmp
mkdir -p gpurun_out/r06
{
python -m pytest tests/test_tracking_gpu.py tests/test_tracking_configs_gpu.py -m gpu -x -q 2>&1 | tail -5
for i in 1 2 3; do
GSH_MC_FAC=0 python profiles/ab/mcorr_ab.py
python profiles/ab/mcorr_ab.py
done
python profiles/ab/mcorr_fixed_cost.py
} > gpurun_out/r06/session2.txt 2>&1
grep -v amdgpu.ids gpurun_out/r06/session2.txt | tail -30
