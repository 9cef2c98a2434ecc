#!/bin/bash
# the final tree: whole GPU suite three times, failures kept in full; smoke
cd /root/repo; mkdir -p gpurun_out/r06
for i in 1 2 3; do
  python -m pytest tests -m gpu -x -q > /tmp/suite_$i.log 2>&1
  echo "suite run $i: $(grep -E ' passed| failed' /tmp/suite_$i.log | tail -1)" >> gpurun_out/r06/session59.txt
  grep -E "repeated once" /tmp/suite_$i.log | cut -c1-600 >> gpurun_out/r06/session59.txt
  if grep -q " failed" /tmp/suite_$i.log; then grep -E "FAIL|churn:|^E  " /tmp/suite_$i.log | cut -c1-3000 | head -60 >> gpurun_out/r06/session59.txt; fi
done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -i "smoke" >> gpurun_out/r06/session59.txt
cat gpurun_out/r06/session59.txt | cut -c1-1500
