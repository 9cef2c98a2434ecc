#!/bin/bash
# (a) the paired trip's floors as v_cvt_pkrtz_f16_f32 over scaled chains (GSH_MC_PKRTZ, build/variants/lib_pk.so) beside the shipped form and the rounding-mode form:
# bit-exactness, then launch times; (b) the churn sixteen more times, failures in full
cd /root/repo; mkdir -p gpurun_out/r06
{
export GSH_LIB_PATH=/root/repo/build/variants/lib_pk.so
timeout 900 python -m pytest tests/test_tracking_gpu.py tests/test_tracking_configs_gpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed|^FAILED|^ERROR|Error" | tail -5
for tag in shipped pk rtn shipped pk rtn shipped pk rtn; do
  if [ $tag = shipped ]; then unset GSH_LIB_PATH; else export GSH_LIB_PATH=/root/repo/build/variants/lib_$tag.so; fi
  echo "== $tag $(timeout 300 python profiles/ab/r06/mcorr_n.py 25000 2>&1 | grep -v amdgpu | tail -1)"
done
unset GSH_LIB_PATH
cd /tmp
for i in $(seq 1 16); do
  timeout 300 /root/repo/tests/host/test_channel churn 32 8 2.4 1 > /tmp/churn_$i.log 2>&1; rc=$?
  echo "run $i rc=$rc $(grep -c FAIL /tmp/churn_$i.log) fails; $(grep -o 'dropped by the time limit' /tmp/churn_$i.log | wc -l) early-drop lines; $(grep -o 'at least [0-9.]* %' /tmp/churn_$i.log)"
  grep FAIL /tmp/churn_$i.log | cut -c1-1200 | head -8
done
} > gpurun_out/r06/session25.txt 2>&1
cat gpurun_out/r06/session25.txt
