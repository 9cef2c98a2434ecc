#!/bin/bash
# Round-5 measurements, part 3: the first round of the cell kernel started in groups (GSH_OC_STAGGER="groups,ticks of 10 ns") x cells per work-group
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05g; mkdir -p $OUT
cd $ROOT
{
for c in 1 3; do
  for st in "0,0" "2,900" "2,600" "3,600" "3,450" "4,450" "4,300" "5,350" "8,220"; do
    echo "cells per work-group $c, stagger $st: $(GSH_OC_CELLS_PER_WG=$c GSH_OC_STAGGER=$st python profiles/ab/acq_ab.py 2>/dev/null)"
  done
done
} > $OUT/acq_stagger.txt 2>&1
cat $OUT/acq_stagger.txt
