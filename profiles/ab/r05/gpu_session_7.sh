#!/bin/bash
# Round-5 measurements, part 7: the shipped defaults (six cells per work-group = 32 persistent work-groups per XCD, next bin prefetched) -- whole acquisition suite, batch times, phases
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05k; mkdir -p $OUT
cd $ROOT
python -m pytest tests/test_acquisition_gpu.py tests/test_acq_two_step_gpu.py tests/test_pcps_detectors_gpu.py tests/test_adapters_gpu.py tests/test_host_classes_gpu.py tests/test_config1_file_input_gpu.py -m gpu -q -x > $OUT/acq_tests.log 2>&1; tail -3 $OUT/acq_tests.log
for c in 1 3 6 8; do echo "cells per work-group $c: $(GSH_OC_CELLS_PER_WG=$c python profiles/ab/acq_ab.py 2>/dev/null)"; done > $OUT/acq_cells.txt 2>&1
echo "defaults: $(python profiles/ab/acq_ab.py 2>/dev/null)" >> $OUT/acq_cells.txt
echo "defaults, no prefetch: $(GSH_OC_PREFETCH=0 python profiles/ab/acq_ab.py 2>/dev/null)" >> $OUT/acq_cells.txt
cat $OUT/acq_cells.txt
GSH_LIB_PATH=$ROOT/build/variants/lib_ocprof.so python profiles/ab/r05/oc_cell_phases.py > $OUT/oc_cell_phases_default.txt 2> $OUT/err.txt
cat $OUT/oc_cell_phases_default.txt
python profiles/acq_scale.py 2>/dev/null | tail -8
