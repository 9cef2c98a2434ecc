#!/bin/bash
# Round-5 measurements, part 28: are the two table waves on the period's critical path?  (timing experiment: they start on stale values at the top of thread 0's section)
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05p; mkdir -p $OUT
{
for rep in 1 2 3; do
for v in current trk_earlysay; do
  if [ $v = current ]; then GSH_LOOP_AB_CONF=lock python profiles/ab/closed_loop_ab.py 2>/dev/null | tail -2; else GSH_LIB_PATH=$ROOT/build/variants/lib_$v.so GSH_LOOP_AB_CONF=lock python profiles/ab/closed_loop_ab.py 2>/dev/null | tail -2; fi
done; done
} > $OUT/closed_loop_early_say.txt 2>&1
cat $OUT/closed_loop_early_say.txt
