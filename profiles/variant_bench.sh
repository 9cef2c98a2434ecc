#!/bin/bash
# kernel-tuning helper (run on the GPU box): time bench.py's tracking kernel for every build/variants/lib_*.so
for f in build/variants/lib_*.so; do
  for rep in 1 2; do
    GSH_LIB_PATH=$PWD/$f python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-acq 2>/dev/null | python -c "
import sys, json
b = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$f', 'kernel_ms=%.4f' % b['roofline']['kernel_ms'], 'Mcorr/s=%.1f' % (b['value'] / 1e6))"
  done
done
