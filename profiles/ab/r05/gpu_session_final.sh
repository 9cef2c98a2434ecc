#!/bin/bash
# Round-5 final measurements: the GPU suite, the bench line (CPU leg included), the rocprofv3 passes of the same command, the bench line again with this round's counters
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05final; mkdir -p $OUT
timeout 2700 python -m pytest tests -m gpu -q > $OUT/gpu_suite.log 2>&1; tail -4 $OUT/gpu_suite.log
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; tail -c 600 $OUT/bench.json
timeout 1500 bash profiles/run_profiles_r05.sh r05 > $OUT/profiles.log 2>&1; tail -5 $OUT/profiles.log
cp profiles/r05_summary.txt profiles/pmc_r05.json $OUT/ 2>/dev/null
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_pmc.json 2> $OUT/bench_pmc.err; tail -c 300 $OUT/bench_pmc.json
