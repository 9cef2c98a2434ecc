#!/bin/bash
# the sub-cells of a decimation-in-time split walked class by class within an XCD (GSH_OC_DIT_R_MAJOR=1) instead of cell by cell
cd /root/repo
mkdir -p gpurun_out/r06
{
for rep in 1 2 3; do
echo "== cell by cell"; GSH_OC_DIT_R_MAJOR=0 python profiles/ab/r06/acq_128k.py 100000:25e6 128000:32e6 200000:50e6 2>&1 | grep "^N ="
echo "== class by class"; GSH_OC_DIT_R_MAJOR=1 python profiles/ab/r06/acq_128k.py 100000:25e6 128000:32e6 200000:50e6 2>&1 | grep "^N ="
done
GSH_OC_DIT_R_MAJOR=1 python -m pytest tests/test_acquisition_gpu.py tests/test_pcps_detectors_gpu.py tests/test_acq_two_step_gpu.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error" | tail -3
} > gpurun_out/r06/session49.txt 2>&1
cat gpurun_out/r06/session49.txt
