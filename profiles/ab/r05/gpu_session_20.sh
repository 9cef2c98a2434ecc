#!/bin/bash
# Round-5 measurements, part 20: the persistent work-groups per XCD and the batches in flight once more, with the idle waves' operand prefetch in
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05p; mkdir -p $OUT
{
for cfg in "2 28 6" "2 24 7" "2 21 8" "2 30 6" "2 32 6" "3 28 6" "3 32 6" "2 28 6"; do
  set -- $cfg
  echo "== $1 lanes, $2 work-groups per XCD, up to $3 cells each"
  GSH_ACQ_LANES=$1 GSH_OC_WG_PER_XCD=$2 GSH_OC_CELLS_PER_WG=$3 timeout 300 python profiles/ab/acq_ab.py 2>&1 | tail -1
done
} > $OUT/acq_retune.txt 2>&1
cat $OUT/acq_retune.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
