#!/bin/bash
# combine launch of the decimation-in-time split WITHOUT the hand-off at the end of a work-group (per-wave records + oc_rows_kernel): parts per cell x threads
cd /root/repo
mkdir -p gpurun_out/r06
{
for rep in 1 2; do
for cfg in "1 1024" "2 1024" "4 1024" "1 512" "2 512" "4 512" "2 768" "1 768" "4 256" "2 256"; do
  set -- $cfg
  GSH_OC_COMBINE_PARTS=$1 GSH_OC_COMBINE_THREADS=$2 python profiles/ab/r06/acq_128k.py 2>&1 | grep "^N ="
done
done
python profiles/ab/r06/acq_128k.py 50000:50e6 100000:25e6 2>&1 | grep "^N ="
for cfg in "1 1024" "2 1024" "4 512"; do
  set -- $cfg
  GSH_OC_COMBINE_PARTS=$1 GSH_OC_COMBINE_THREADS=$2 python -m pytest tests/test_acquisition_gpu.py tests/test_pcps_detectors_gpu.py tests/test_acq_two_step_gpu.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error" | tail -3
done
} > gpurun_out/r06/session42.txt 2>&1
cat gpurun_out/r06/session42.txt
