#!/bin/bash
# Round-5 measurements, part 23: the arg-max's index formed once per thread instead of once per element
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05p; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q -x -k "acq or pcps or onchip" > $OUT/acq_tests_kbest.log 2>&1; tail -3 $OUT/acq_tests_kbest.log
{
for rep in 1 2 3; do
  GSH_X=0 timeout 300 python profiles/ab/acq_ab.py 2>&1 | tail -1
done
} > $OUT/acq_kbest.txt 2>&1
cat $OUT/acq_kbest.txt
