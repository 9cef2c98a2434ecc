#!/bin/bash
# round 6, session 10: decimation-in-time sub-cells on the persistent cell kernel (GSH_OC_DIT_PERSIST=0: as before); N = 50 000 on DIT as well (GSH_OC_DIT_MIN_S=2)
cd /root/repo
mkdir -p gpurun_out/r06
{
python -m pytest tests/test_acquisition_gpu.py -m gpu -x -q -k "split or 128000 or long or dit" 2>&1 | tail -3
for i in 1 2; do
echo "== persistent sub-cells"; python profiles/ab/r06/acq_split_scale.py 2>&1 | grep " 32 PRN"
echo "== one work-group per sub-cell"; GSH_OC_DIT_PERSIST=0 python profiles/ab/r06/acq_split_scale.py 2>&1 | grep " 32 PRN"
done
echo "== DIT from S = 2"; GSH_OC_DIT_MIN_S=2 python profiles/ab/r06/acq_split_scale.py 2>&1 | grep " 32 PRN"
echo "== DIT from S = 2, one work-group per sub-cell"; GSH_OC_DIT_PERSIST=0 GSH_OC_DIT_MIN_S=2 python profiles/ab/r06/acq_split_scale.py 2>&1 | grep " 32 PRN"
} > gpurun_out/r06/session10.txt 2>&1
grep -v amdgpu.ids gpurun_out/r06/session10.txt | tail -40
