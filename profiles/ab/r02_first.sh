#!/bin/bash
# round-2 first GPU pass: GPU parity suite on the packed correlator body + LDS-resident closed-loop state, then A/B timings
mkdir -p gpurun_out/r02a
python -m pytest tests -m gpu -x -q > gpurun_out/r02a/gpu_suite.log 2>&1
tail -5 gpurun_out/r02a/gpu_suite.log
for v in 0 1 0 1; do GSH_MC_PACKED_BODY=$v python profiles/ab/mcorr_ab.py 2>&1 | tail -1 | sed "s/^/packed=$v /"; done | tee gpurun_out/r02a/mcorr_ab.log
python profiles/ab/closed_loop_ab.py 2>&1 | tail -2 | tee gpurun_out/r02a/closed_loop_ab.log
for v in 0 1; do GSH_MC_PACKED_BODY=$v python profiles/config_rates.py 2>&1 | tail -6 | sed "s/^/packed=$v /"; done | tee gpurun_out/r02a/config_rates.log
