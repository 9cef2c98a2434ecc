#!/bin/bash
# Round-5 measurements, part 11: the closed-loop kernel compiled without MachineLICM (fewer hoisted scalars: 98 -> 71 spilled SGPRs, 128 -> 106 VGPRs) and with other
# scheduling strategies; 32 and 256 channels, lock detectors on
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05o; mkdir -p $OUT
cd $ROOT
{
for v in current trk_nolicm trk_nolicm_maxilp trk_maxilp current trk_nolicm; do
  if [ $v = current ]; then GSH_LOOP_AB_CONF=lock python profiles/ab/closed_loop_ab.py 2>/dev/null; else GSH_LIB_PATH=$ROOT/build/variants/lib_$v.so GSH_LOOP_AB_CONF=lock python profiles/ab/closed_loop_ab.py 2>/dev/null; fi
done
for v in current trk_nolicm; do
  if [ $v = current ]; then GSH_LOOP_AB_CONF= GSH_LOOP_AB_CH=32 python profiles/ab/closed_loop_ab.py 2>/dev/null; else GSH_LIB_PATH=$ROOT/build/variants/lib_$v.so GSH_LOOP_AB_CONF= GSH_LOOP_AB_CH=32 python profiles/ab/closed_loop_ab.py 2>/dev/null; fi
done
} > $OUT/closed_loop_flags_ab.txt 2>&1
cat $OUT/closed_loop_flags_ab.txt
