#!/bin/bash
# shipped = non-temporal Z + ordered lanes on decimation-in-time plans; against free-running lanes and against the library without the hints, same box
cd /root/repo
mkdir -p gpurun_out/r06
{
for rep in 1 2 3; do
  echo "== shipped (NT, ordered)"; python profiles/ab/r06/acq_128k.py 2>&1 | grep "^N ="
  echo "== NT, free lanes"; GSH_ACQ_DIT_ORDERED=0 python profiles/ab/r06/acq_128k.py 2>&1 | grep "^N ="
  echo "== no NT, ordered"; GSH_LIB_PATH=/root/repo/build/variants/lib_nont.so python profiles/ab/r06/acq_128k.py 2>&1 | grep "^N ="
  echo "== no NT, free lanes"; GSH_ACQ_DIT_ORDERED=0 GSH_LIB_PATH=/root/repo/build/variants/lib_nont.so python profiles/ab/r06/acq_128k.py 2>&1 | grep "^N ="
done
echo "== shipped, other split shapes"; python profiles/ab/r06/acq_128k.py 50000:50e6 100000:25e6 160000:40e6 200000:50e6 2>&1 | grep "^N ="
echo "== no NT, free lanes, other split shapes"; GSH_ACQ_DIT_ORDERED=0 GSH_LIB_PATH=/root/repo/build/variants/lib_nont.so python profiles/ab/r06/acq_128k.py 50000:50e6 100000:25e6 160000:40e6 200000:50e6 2>&1 | grep "^N ="
python -m pytest tests/test_acquisition_gpu.py tests/test_pcps_detectors_gpu.py tests/test_acq_two_step_gpu.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error" | tail -3
} > gpurun_out/r06/session44.txt 2>&1
cat gpurun_out/r06/session44.txt
