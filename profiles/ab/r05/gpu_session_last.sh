#!/bin/bash
# the round's last tree: the GPU suite and the bench line
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05final; mkdir -p $OUT
timeout 2700 python -m pytest tests -m gpu -q > $OUT/gpu_suite_last.log 2>&1; tail -2 $OUT/gpu_suite_last.log
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_last.json 2> $OUT/bench_last.err
echo "stdout lines: $(wc -l < $OUT/bench_last.json)"; tail -c 600 $OUT/bench_last.json
