#!/bin/bash
# Round-5 measurements, part 24: closed loop, the lanes' seeds from tables filled by two idle waves (shipped) against one evaluation per lane (-DGSH_TRK_SEED_TABLES=0)
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05p; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -x -k "loop or live or symbol or trk" > $OUT/loop_tests_seeds.log 2>&1; tail -5 $OUT/loop_tests_seeds.log
{
for rep in 1 2 3; do
for v in current trk_noseeds; do
  if [ $v = current ]; then GSH_LOOP_AB_CONF=lock python profiles/ab/closed_loop_ab.py 2>/dev/null | tail -2; else GSH_LIB_PATH=$ROOT/build/variants/lib_$v.so GSH_LOOP_AB_CONF=lock python profiles/ab/closed_loop_ab.py 2>/dev/null | tail -2; fi
done; done
for v in current trk_noseeds; do
  if [ $v = current ]; then GSH_LOOP_AB_CONF= python profiles/ab/closed_loop_ab.py 2>/dev/null | tail -2; else GSH_LIB_PATH=$ROOT/build/variants/lib_$v.so GSH_LOOP_AB_CONF= python profiles/ab/closed_loop_ab.py 2>/dev/null | tail -2; fi
done
} > $OUT/closed_loop_seed_tables.txt 2>&1
cat $OUT/closed_loop_seed_tables.txt
