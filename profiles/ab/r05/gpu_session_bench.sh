#!/bin/bash
# the bench line of the tree as it stands (stdout must carry the JSON line only; stderr the rest)
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05final; mkdir -p $OUT
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_last.json 2> $OUT/bench_last.err
echo "stdout lines: $(wc -l < $OUT/bench_last.json)"; tail -c 700 $OUT/bench_last.json
