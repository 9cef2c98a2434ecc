#!/bin/bash
# the acquisition tests and the batch time with exchange 2 writing first (shipped from here on)
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05p; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q -x -k "acq or pcps or onchip or detector or tong or quicksync or e5a" > $OUT/acq_tests_wfirst.log 2>&1; tail -3 $OUT/acq_tests_wfirst.log
timeout 300 python profiles/ab/acq_ab.py 2>&1 | tail -1
