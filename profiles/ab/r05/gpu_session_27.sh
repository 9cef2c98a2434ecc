#!/bin/bash
# Round-5 measurements, part 27: config 4 closed loop (re-seeded windows), the lane rotation exp(-j 2 tid step) from the seed tables (shipped) against evaluated per lane
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05p; mkdir -p $OUT
cat > /tmp/c4.py <<'P'
import sys, os, json
sys.path.insert(0, os.getcwd())
import torch, bench
d = bench.closed_loop_config4_metric(torch, 0)
print(os.environ.get("GSH_LIB_PATH", "shipped"), {k: d[k] for k in ("us_per_epoch", "value", "channels_with_signal_locked")})
P
{
for rep in 1 2; do
python /tmp/c4.py 2>/dev/null | tail -1
GSH_LIB_PATH=$ROOT/build/variants/lib_trk_before_lf.so python /tmp/c4.py 2>/dev/null | tail -1
done
} > $OUT/config4_lf.txt 2>&1
cat $OUT/config4_lf.txt
timeout 1500 python -m pytest tests -m gpu -q -x -k "loop or live or symbol or trk" > $OUT/loop_tests_lf.log 2>&1; tail -3 $OUT/loop_tests_lf.log
