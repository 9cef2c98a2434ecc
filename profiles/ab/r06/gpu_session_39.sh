#!/bin/bash
# the 2 040 bound of the paired-tap judgement only where two floors per instruction are used (whole-code tables): the windowed tables of long codes pair their taps up
# to 2^16 again.  Tracking tests, then configs 4 / 5 with the library before the fix (build/variants/lib_nomid.so = the previous commit's code) and after
cd /root/repo; mkdir -p gpurun_out/r06
{
timeout 900 python -m pytest tests/test_tracking_gpu.py tests/test_tracking_configs_gpu.py tests/test_host_classes_gpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed|^FAILED|^ERROR|Error" | tail -5
for tag in nomid shipped nomid shipped nomid shipped; do
  if [ $tag = shipped ]; then unset GSH_LIB_PATH; else export GSH_LIB_PATH=/root/repo/build/variants/lib_$tag.so; fi
  echo "== $tag"; timeout 300 python profiles/config_rates.py 2>&1 | grep -v amdgpu | tail -4
done
} > gpurun_out/r06/session39.txt 2>&1
cat gpurun_out/r06/session39.txt
