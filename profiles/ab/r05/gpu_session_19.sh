#!/bin/bash
# Round-5 measurements, part 19: closed loop, the first trip's loads issued ahead of the seed evaluation (shipped) against right in front of the first trip (-DGSH_MC_EARLY_LOADS=0)
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05p; mkdir -p $OUT
{
for rep in 1 2 3; do
for v in current trk_late; do
  if [ $v = current ]; then GSH_LOOP_AB_CONF=lock python profiles/ab/closed_loop_ab.py 2>/dev/null | tail -2; else GSH_LIB_PATH=$ROOT/build/variants/lib_$v.so GSH_LOOP_AB_CONF=lock python profiles/ab/closed_loop_ab.py 2>/dev/null | tail -2; fi
done; done
for v in current trk_late; do
  if [ $v = current ]; then GSH_LOOP_AB_CONF= python profiles/ab/closed_loop_ab.py 2>/dev/null | tail -2; else GSH_LIB_PATH=$ROOT/build/variants/lib_$v.so GSH_LOOP_AB_CONF= python profiles/ab/closed_loop_ab.py 2>/dev/null | tail -2; fi
done
} > $OUT/closed_loop_early_loads.txt 2>&1
cat $OUT/closed_loop_early_loads.txt
timeout 1200 python -m pytest tests -m gpu -q -x -k "loop or live or symbol or trk" > $OUT/loop_tests.log 2>&1; tail -3 $OUT/loop_tests.log
