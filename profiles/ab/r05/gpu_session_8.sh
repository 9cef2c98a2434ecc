#!/bin/bash
# Round-5 measurements, part 8: persistent work-groups per XCD (GSH_OC_WG_PER_XCD) with six cells per work-group and the next bin prefetched
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05l; mkdir -p $OUT
cd $ROOT
for w in 20 24 26 28 30 32; do echo "work-groups per XCD $w: $(GSH_OC_WG_PER_XCD=$w python profiles/ab/acq_ab.py 2>/dev/null)"; done > $OUT/acq_wg_per_xcd.txt 2>&1
cat $OUT/acq_wg_per_xcd.txt
python -m pytest tests/test_acquisition_gpu.py tests/test_acq_two_step_gpu.py tests/test_pcps_detectors_gpu.py tests/test_adapters_gpu.py tests/test_host_classes_gpu.py tests/test_config1_file_input_gpu.py -m gpu -q -x > $OUT/acq_tests.log 2>&1; tail -3 $OUT/acq_tests.log
python profiles/acq_scale.py 2>/dev/null | tail -9 > $OUT/acq_scale.txt; cat $OUT/acq_scale.txt
