#!/bin/bash
# Round-5 measurements, part 13: how the batch time depends on the number of persistent work-groups per XCD (cells dealt by ticket, so any number balances)
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05p; mkdir -p $OUT
{
for w in 4 8 12 16 20 24 28; do
  echo "== tickets 1, $w work-groups per XCD"
  GSH_OC_TICKETS=1 GSH_OC_WG_PER_XCD=$w timeout 300 python profiles/ab/acq_ab.py 2>&1 | tail -1
done
for w in 16 20 24; do
  echo "== tickets 1, $w work-groups per XCD, no prefetch"
  GSH_OC_PREFETCH=0 GSH_OC_TICKETS=1 GSH_OC_WG_PER_XCD=$w timeout 300 python profiles/ab/acq_ab.py 2>&1 | tail -1
done
} > $OUT/acq_wg_sweep.txt 2>&1
cat $OUT/acq_wg_sweep.txt
