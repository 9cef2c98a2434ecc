#!/bin/bash
# Round-5 measurements, part 26: thread 0 sets seed_ok without reading back what it said (shipped) against the read-back
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05p; mkdir -p $OUT
{
for rep in 1 2 3; do
for v in current trk_readback; do
  if [ $v = current ]; then GSH_LOOP_AB_CONF=lock python profiles/ab/closed_loop_ab.py 2>/dev/null | tail -2; else GSH_LIB_PATH=$ROOT/build/variants/lib_$v.so GSH_LOOP_AB_CONF=lock python profiles/ab/closed_loop_ab.py 2>/dev/null | tail -2; fi
done; done
} > $OUT/closed_loop_seed_reads.txt 2>&1
cat $OUT/closed_loop_seed_reads.txt
timeout 1500 python -m pytest tests -m gpu -q -x -k "loop or live or symbol or trk or corr" > $OUT/loop_tests_seeds2.log 2>&1; tail -3 $OUT/loop_tests_seeds2.log
