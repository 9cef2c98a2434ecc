#!/bin/bash
R=$(pwd); mkdir -p gpurun_out/r02b
python -m pytest tests -m gpu -x -q > gpurun_out/r02b/gpu_suite.log 2>&1; tail -3 gpurun_out/r02b/gpu_suite.log
for rep in 1 2; do
  for v in 0 1; do GSH_MC_PACKED_BODY=$v python profiles/ab/mcorr_ab.py 2>&1 | tail -1 | sed "s/^/t256 packed=$v /"; done
  for f in build/variants/lib_*.so; do GSH_LIB_PATH=$R/$f python profiles/ab/mcorr_ab.py 2>&1 | tail -1; done
done | tee gpurun_out/r02b/mcorr_ab.log
python profiles/ab/closed_loop_ab.py 2>&1 | tail -2 | tee gpurun_out/r02b/closed_loop_ab.log
python profiles/config_rates.py 2>&1 | grep "splits  0\|config 4" | tee gpurun_out/r02b/config_rates.log
