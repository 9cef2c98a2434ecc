#!/bin/bash
# Round-5 measurements, part 22: no barrier between the passes of the cell kernel (shipped) against one (-DGSH_OC_PASS_BARRIER=1)
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05p; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q -x -k "acq or pcps or onchip" > $OUT/acq_tests_nobar.log 2>&1; tail -3 $OUT/acq_tests_nobar.log
{
for rep in 1 2 3; do
for lib in "" build/variants/lib_passbar.so; do
  echo "== ${lib:-shipped (no pass barrier)}"
  GSH_LIB_PATH=${lib:+$ROOT/$lib} timeout 300 python profiles/ab/acq_ab.py 2>&1 | tail -1
done; done
} > $OUT/acq_pass_barrier.txt 2>&1
cat $OUT/acq_pass_barrier.txt
