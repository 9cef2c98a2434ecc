#!/bin/bash
# Z of the decimation-in-time split: non-temporal stores (sub-cells) / loads (combine), against the shipped library on the same box
cd /root/repo
mkdir -p gpurun_out/r06
{
for rep in 1 2 3; do
for v in shipped nts ntl ntsl; do
  if [ $v = shipped ]; then unset GSH_LIB_PATH; else export GSH_LIB_PATH=/root/repo/build/variants/lib_$v.so; fi
  echo "== $v"; python profiles/ab/r06/acq_128k.py 2>&1 | grep "^N ="
done
done
unset GSH_LIB_PATH
python -m pytest tests/test_acquisition_gpu.py tests/test_pcps_detectors_gpu.py tests/test_acq_two_step_gpu.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error" | tail -3
} > gpurun_out/r06/session43.txt 2>&1
cat gpurun_out/r06/session43.txt
