#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r06
{
python -m pytest tests/test_acquisition_gpu.py tests/test_pcps_detectors_gpu.py tests/test_acq_two_step_gpu.py -m gpu -x -q 2>&1 | tail -3
for i in 1 2; do
echo "== paired combine"; python profiles/ab/r06/acq_split_scale.py 2>&1 | grep " 32 PRN"
echo "== paired combine, DIT from S = 2"; GSH_OC_DIT_MIN_S=2 python profiles/ab/r06/acq_split_scale.py 2>&1 | grep " 32 PRN"
done
} > gpurun_out/r06/session12.txt 2>&1
grep -v amdgpu.ids gpurun_out/r06/session12.txt | tail -40
