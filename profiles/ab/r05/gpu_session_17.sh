#!/bin/bash
# Round-5 measurements, part 17: the idle waves of stage 3 load their own operands of the next cell (GSH_OC_PREFETCH=2, the build's default) against touching the
# next bin's lines only (=1)
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05p; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q -x -k "acq or pcps or onchip" > $OUT/acq_tests_opf.log 2>&1; tail -3 $OUT/acq_tests_opf.log
{
for rep in 1 2; do
for pf in 1 2; do
for cfg in "2 28" "3 32"; do
  set -- $cfg
  echo "== prefetch $pf, $1 lanes, $2 work-groups per XCD"
  GSH_OC_PREFETCH=$pf GSH_ACQ_LANES=$1 GSH_OC_WG_PER_XCD=$2 timeout 300 python profiles/ab/acq_ab.py 2>&1 | tail -1
done; done; done
} > $OUT/acq_opf.txt 2>&1
cat $OUT/acq_opf.txt
