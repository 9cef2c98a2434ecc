#!/bin/bash
# Round-5 measurements, part 4: cells per work-group with the registers declared fresh by empty asm (no zero fill in front of the operand loads)
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05h; mkdir -p $OUT
cd $ROOT
for c in 1 2 3 4 6; do echo "cells per work-group $c: $(GSH_OC_CELLS_PER_WG=$c python profiles/ab/acq_ab.py 2>/dev/null)"; done > $OUT/acq_cells_per_wg.txt 2>&1
cat $OUT/acq_cells_per_wg.txt
for c in 1 3; do GSH_OC_CELLS_PER_WG=$c GSH_LIB_PATH=$ROOT/build/variants/lib_ocprof.so python profiles/ab/r05/oc_cell_phases.py > $OUT/oc_cell_phases_cpw$c.txt 2> $OUT/oc_cell_phases_cpw$c.err; done
cat $OUT/oc_cell_phases_cpw1.txt; grep -A12 "clocks per stage" $OUT/oc_cell_phases_cpw3.txt; grep "this run\|time line" $OUT/oc_cell_phases_cpw3.txt
GSH_OC_CELLS_PER_WG=3 python -m pytest tests/test_acquisition_gpu.py tests/test_acq_two_step_gpu.py tests/test_pcps_detectors_gpu.py tests/test_adapters_gpu.py -m gpu -q -x > $OUT/acq_tests_cpw3.log 2>&1; tail -3 $OUT/acq_tests_cpw3.log
