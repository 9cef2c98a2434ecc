#!/bin/bash
# whole GPU suite three times, every failure's full text kept (the channel churn failed once in session 50's run and not in ten runs on its own)
cd /root/repo; mkdir -p gpurun_out/r06
for i in 1 2 3; do
  python -m pytest tests -m gpu -x -q > /tmp/suite_$i.log 2>&1
  echo "suite run $i: $(grep -E ' passed| failed' /tmp/suite_$i.log | tail -1)" >> gpurun_out/r06/session52.txt
  if grep -q " failed" /tmp/suite_$i.log; then grep -E "FAIL|churn:|Error|assert" /tmp/suite_$i.log | cut -c1-3000 | head -60 >> gpurun_out/r06/session52.txt; fi
done
cat gpurun_out/r06/session52.txt | cut -c1-1500
