#!/bin/bash
# (a) the GPU suite and six churn runs after the symbol-count allowance follows the receivers' hand-over times; (b) the paired trip's floor() as a packed fma under
# round-toward-minus-infinity (GSH_MC_RTN_FLOOR, build/variants/lib_rtn.so) beside the shipped v_cvt_flr_i32_f32: bit-exactness, then the launch time
cd /root/repo; mkdir -p gpurun_out/r06
{
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|^FAILED|^ERROR" | tail -5
cd /tmp
for i in 1 2 3 4 5 6; do
  timeout 300 /root/repo/tests/host/test_channel churn 32 8 2.4 1 > /tmp/churn_$i.log 2>&1; rc=$?
  echo "run $i rc=$rc $(grep -c FAIL /tmp/churn_$i.log) fails; $(grep -o 'dropped by the time limit' /tmp/churn_$i.log | wc -l) early-drop lines; $(grep -o 'at least [0-9.]* %' /tmp/churn_$i.log)"
  grep FAIL /tmp/churn_$i.log | cut -c1-600 | head -5
done
cd /root/repo
export GSH_LIB_PATH=/root/repo/build/variants/lib_rtn.so
timeout 900 python -m pytest tests/test_tracking_gpu.py tests/test_tracking_configs_gpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed|^FAILED|^ERROR|Error" | tail -5
for tag in shipped rtn shipped rtn shipped rtn; do
  if [ $tag = shipped ]; then unset GSH_LIB_PATH; else export GSH_LIB_PATH=/root/repo/build/variants/lib_$tag.so; fi
  echo "== $tag"; timeout 300 python profiles/ab/r06/mcorr_n.py 25000 2>&1 | grep -v amdgpu | tail -1
done
} > gpurun_out/r06/session23.txt 2>&1
cat gpurun_out/r06/session23.txt
