#!/bin/bash
# Round-5 measurements, part 2 (on the GPU box, from the repo root): the acquisition cells without the in-kernel hand-off (oc_rows_kernel) and with several cells per
# work-group (GSH_OC_CELLS_PER_WG): parity tests, batch times, stage clocks.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05f; mkdir -p $OUT
cd $ROOT
python -m pytest tests/test_acquisition_gpu.py tests/test_acq_two_step_gpu.py tests/test_pcps_detectors_gpu.py tests/test_adapters_gpu.py -m gpu -q -x > $OUT/acq_tests_cpw1.log 2>&1; tail -3 $OUT/acq_tests_cpw1.log
for c in 1 2 3 6; do echo "cells per work-group $c: $(GSH_OC_CELLS_PER_WG=$c python profiles/ab/acq_ab.py 2>/dev/null)"; done > $OUT/acq_cells_per_wg.txt 2>&1
cat $OUT/acq_cells_per_wg.txt
GSH_OC_CELLS_PER_WG=3 python -m pytest tests/test_acquisition_gpu.py tests/test_acq_two_step_gpu.py tests/test_pcps_detectors_gpu.py -m gpu -q -x > $OUT/acq_tests_cpw3.log 2>&1; tail -3 $OUT/acq_tests_cpw3.log
GSH_OC_CELLS_PER_WG=6 python -m pytest tests/test_acquisition_gpu.py -m gpu -q -x > $OUT/acq_tests_cpw6.log 2>&1; tail -3 $OUT/acq_tests_cpw6.log
for c in 1 6; do GSH_OC_CELLS_PER_WG=$c GSH_LIB_PATH=$ROOT/build/variants/lib_ocprof.so python profiles/ab/r05/oc_cell_phases.py > $OUT/oc_cell_phases_cpw$c.txt 2> $OUT/oc_cell_phases_cpw$c.err; done
cat $OUT/oc_cell_phases_cpw1.txt $OUT/oc_cell_phases_cpw6.txt
