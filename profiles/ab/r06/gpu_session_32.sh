#!/bin/bash
# closed loop: look-ups with the table's absolute address in the instruction (the shipped library) against v_lshl_add_u32 per look-up (build/variants/lib_nokc.so);
# records must be byte-identical; then the closed-loop tests
cd /root/repo; mkdir -p gpurun_out/r06
{
for tag in nokc shipped nokc shipped nokc shipped; do
  if [ $tag = shipped ]; then unset GSH_LIB_PATH; else export GSH_LIB_PATH=/root/repo/build/variants/lib_$tag.so; fi
  echo "== $tag"; timeout 300 python profiles/ab/r06/closed_loop_kc.py 2>&1 | grep -v amdgpu | tail -2
done
unset GSH_LIB_PATH
timeout 1500 python -m pytest tests/test_tracking_loop_gpu.py tests/test_symbol_sync.py tests/test_trk_dump.py tests/test_tracking_adapters.py -x -q -m gpu > /tmp/t.log 2>&1
grep -E "FAIL|passed|failed|^ERROR|Error" /tmp/t.log | cut -c1-800 | tail -8
} > gpurun_out/r06/session32.txt 2>&1
cat gpurun_out/r06/session32.txt
