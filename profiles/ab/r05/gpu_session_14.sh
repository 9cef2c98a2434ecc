#!/bin/bash
# Round-5 measurements, part 14: two, three and four batches in flight (GSH_ACQ_LANES) on 28 / 30 / 32 persistent work-groups per XCD, cells by ticket or six each
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05p; mkdir -p $OUT
{
for lanes in 2 3 4; do
for tk in 1 0; do
for w in 28 30 32; do
  echo "== $lanes lanes, tickets $tk, $w work-groups per XCD"
  GSH_ACQ_LANES=$lanes GSH_OC_TICKETS=$tk GSH_OC_WG_PER_XCD=$w timeout 300 python profiles/ab/acq_ab.py 2>&1 | tail -1
done; done; done
} > $OUT/acq_lanes.txt 2>&1
cat $OUT/acq_lanes.txt
