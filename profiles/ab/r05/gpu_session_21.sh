#!/bin/bash
# Round-5 measurements, part 21: stage clocks of the cell with the idle waves' operand prefetch, wave by wave
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05p; mkdir -p $OUT
GSH_OC_CELLS_PER_WG=6 GSH_LIB_PATH=$ROOT/build/variants/lib_ocprof.so python profiles/ab/r05/oc_cell_phases.py > $OUT/oc_cell_phases_opf.txt 2>&1
head -30 $OUT/oc_cell_phases_opf.txt | cut -c100-200; tail -20 $OUT/oc_cell_phases_opf.txt
GSH_OC_PREFETCH=1 GSH_OC_CELLS_PER_WG=6 GSH_LIB_PATH=$ROOT/build/variants/lib_ocprof.so python profiles/ab/r05/oc_cell_phases.py > $OUT/oc_cell_phases_pf1.txt 2>&1
tail -18 $OUT/oc_cell_phases_pf1.txt
