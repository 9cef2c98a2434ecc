#!/bin/bash
# the GPU suite of the tree as it stands
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05final; mkdir -p $OUT
timeout 2700 python -m pytest tests -m gpu -q > $OUT/gpu_suite_last.log 2>&1; tail -3 $OUT/gpu_suite_last.log
