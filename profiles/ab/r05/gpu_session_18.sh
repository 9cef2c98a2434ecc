#!/bin/bash
# Round-5 measurements, part 18: BASELINE config 4 closed loop (E1, 5 + 1 taps) with two chunks per trip in the five-tap / pilot + data flavours (-DGSH_MC_NCH_WIDE) against one
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05p; mkdir -p $OUT
cat > /tmp/c4.py <<'P'
import sys, os, json
sys.path.insert(0, os.getcwd())
import torch, bench
d = bench.closed_loop_config4_metric(torch, 0)
print(os.environ.get("GSH_LIB_PATH", "shipped"), {k: d[k] for k in d if not isinstance(d[k], (dict, list))})
P
{
for rep in 1 2; do
python /tmp/c4.py 2>/dev/null | tail -1
GSH_LIB_PATH=$ROOT/build/variants/lib_trk_nchwide.so python /tmp/c4.py 2>/dev/null | tail -1
done
} > $OUT/config4_nch.txt 2>&1
cat $OUT/config4_nch.txt
