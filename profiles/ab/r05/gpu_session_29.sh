#!/bin/bash
# Round-5 measurements, part 29: the stage order (25, 40, 25) against (25, 25, 40), both WITHOUT the idle waves' prefetch (the order (25, 40, 25) has no idle wave in stage 3;
# GSH_OC_PREFETCH=0 for both) -- what the issue-bound stages gain from the ten-wave stage carrying no |.|^2 / arg-max
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05p; mkdir -p $OUT
{
for rep in 1 2 3; do
  echo "== (25, 25, 40), no prefetch";  GSH_OC_PREFETCH=0 timeout 300 python profiles/ab/acq_ab.py 2>&1 | tail -1
  echo "== (25, 40, 25), no prefetch";  GSH_OC_PREFETCH=0 GSH_LIB_PATH=$ROOT/build/variants/lib_p254025.so timeout 300 python profiles/ab/acq_ab.py 2>&1 | tail -1
done
echo "== (25, 25, 40), shipped (prefetch 2)"; timeout 300 python profiles/ab/acq_ab.py 2>&1 | tail -1
} > $OUT/acq_stage_order.txt 2>&1
cat $OUT/acq_stage_order.txt
