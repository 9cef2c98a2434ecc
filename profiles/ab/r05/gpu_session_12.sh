#!/bin/bash
# Round-5 measurements, part 12: the cells of a launch dealt to its work-groups by ticket (GSH_OC_TICKETS=1, the build's default) against six each (=0)
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05p; mkdir -p $OUT
{
for rep in 1 2; do
for tk in 0 1; do
for w in 28 30 32; do
  echo "== tickets $tk, $w work-groups per XCD"
  GSH_OC_TICKETS=$tk GSH_OC_WG_PER_XCD=$w timeout 300 python profiles/ab/acq_ab.py 2>&1 | tail -1
done; done; done
echo "== tickets 1, 24 / 26 work-groups per XCD"
GSH_OC_TICKETS=1 GSH_OC_WG_PER_XCD=24 timeout 300 python profiles/ab/acq_ab.py 2>&1 | tail -1
GSH_OC_TICKETS=1 GSH_OC_WG_PER_XCD=26 timeout 300 python profiles/ab/acq_ab.py 2>&1 | tail -1
} > $OUT/acq_tickets.txt 2>&1
cat $OUT/acq_tickets.txt
timeout 900 python -m pytest tests -m gpu -q -x -k "acq or pcps or onchip" > $OUT/acq_tests.log 2>&1; tail -3 $OUT/acq_tests.log
