#!/bin/bash
# Round-5 measurements, part 6: the next cell's bin spectrum touched by the threads that idle through stage 3 (GSH_OC_PREFETCH) x cells per work-group
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05j; mkdir -p $OUT
cd $ROOT
for pf in 0 1; do for c in 1 2 3 4 6; do echo "prefetch $pf, cells per work-group $c: $(GSH_OC_PREFETCH=$pf GSH_OC_CELLS_PER_WG=$c python profiles/ab/acq_ab.py 2>/dev/null)"; done; done > $OUT/acq_prefetch.txt 2>&1
cat $OUT/acq_prefetch.txt
GSH_OC_CELLS_PER_WG=3 GSH_LIB_PATH=$ROOT/build/variants/lib_ocprof.so python profiles/ab/r05/oc_cell_phases.py > $OUT/oc_cell_phases_cpw3_prefetch.txt 2> $OUT/err.txt
grep "this run" $OUT/oc_cell_phases_cpw3_prefetch.txt; grep -A9 "^stage" $OUT/oc_cell_phases_cpw3_prefetch.txt | cut -c1-60,118-170
GSH_OC_CELLS_PER_WG=3 python -m pytest tests/test_acquisition_gpu.py tests/test_acq_two_step_gpu.py tests/test_pcps_detectors_gpu.py tests/test_adapters_gpu.py -m gpu -q -x > $OUT/acq_tests_cpw3.log 2>&1; tail -3 $OUT/acq_tests_cpw3.log
