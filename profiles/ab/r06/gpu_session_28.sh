#!/bin/bash
# how often does the GPU suite fail?  six runs of the whole suite on one box, every failure in full
cd /root/repo; mkdir -p gpurun_out/r06
{
for i in 1 2 3 4 5 6; do
  timeout 1500 python -m pytest tests -x -q -m gpu > /tmp/suite_$i.log 2>&1
  echo "suite run $i: $(grep -E ' passed| failed' /tmp/suite_$i.log | tail -1)"
  grep -E "^FAILED|^ERROR|FAIL " /tmp/suite_$i.log | cut -c1-1500 | head -12
done
} > gpurun_out/r06/session28.txt 2>&1
cat gpurun_out/r06/session28.txt
