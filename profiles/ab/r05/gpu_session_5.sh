#!/bin/bash
# Round-5 measurements, part 5: where do the 7.4 us of the operand phase go?  Every cell reads the spectrum of bin 0 (one 200 KB spectrum for the whole chip: L2 hits after the
# first touch) or of bin & 7 (eight spectra = what one XCD has in flight in a round, never replaced)
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05i; mkdir -p $OUT
cd $ROOT
for v in ocprof ocprof_bin0 ocprof_bin7; do
  GSH_OC_CELLS_PER_WG=1 GSH_LIB_PATH=$ROOT/build/variants/lib_$v.so python profiles/ab/r05/oc_cell_phases.py > $OUT/phases_$v.txt 2> $OUT/phases_$v.err
  echo "== $v"; grep "this run" $OUT/phases_$v.txt; grep -A9 "^stage" $OUT/phases_$v.txt | cut -c1-60,118-170
done
