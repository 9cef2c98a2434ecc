#!/bin/bash
# the XCD tile: 8 / 4 / 2 XCDs across the PRNs (4 PRNs x 41 bins, 8 x 21, 16 x 11 per XCD) at 25 000, 50 000, 128 000 points
cd /root/repo; mkdir -p gpurun_out/r06
{
for rep in 1 2; do
for xp in 8 4 2; do
  echo "== GSH_OC_XP=$xp"; GSH_OC_XP=$xp python profiles/ab/r06/acq_128k.py 25000:25e6 50000:50e6 128000:32e6 2>&1 | grep "^N ="
done; done
GSH_OC_XP=4 python -m pytest tests/test_acquisition_gpu.py tests/test_pcps_detectors_gpu.py tests/test_acq_two_step_gpu.py -m gpu -x -q 2>&1 | grep -E " passed| failed" | tail -1
} > gpurun_out/r06/session61.txt 2>&1
cat gpurun_out/r06/session61.txt
