#!/bin/bash
# timing split of the run-based correlator path (GSH_MC_PACKED_BODY=2): full, without the boundary phase, without prefix + boundary phases
cd "$(dirname "$0")/../.."
R=$(pwd)
GSH_MC_PACKED_BODY=1 python profiles/ab/mcorr_ab.py 2>&1 | tail -1 | sed "s/^/packed trips: /"
for f in build/variants/lib_*.so; do
  GSH_MC_PACKED_BODY=2 GSH_LIB_PATH=$R/$f python profiles/ab/mcorr_ab.py 2>&1 | tail -1
done
echo "--- parity with the run-based path forced (standard-mode jobs that qualify take it)"
GSH_MC_PACKED_BODY=2 GSH_LIB_PATH=$R/build/variants/lib_full.so timeout 900 python -m pytest tests/test_tracking_gpu.py tests/test_tracking_configs_gpu.py tests/test_sample_stream_gpu.py -m gpu -q 2>&1 | grep -E "^FAILED|passed|failed|Error|fault" | head -20
