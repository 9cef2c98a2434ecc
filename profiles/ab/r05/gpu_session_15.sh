#!/bin/bash
# Round-5 measurements, part 15: what bounds the batch at ~100 us whatever the packing?  (a) every cell reads bin 0 (L2 hits only), (b) no operand loads at all,
# (c) the clocks and the power while batches run for seconds
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05p; mkdir -p $OUT
{
for lib in "" build/variants/lib_bin0.so build/variants/lib_noloads.so; do
for cfg in "2 28" "3 32" "4 32"; do
  set -- $cfg
  echo "== lib ${lib:-shipped}, $1 lanes, $2 work-groups per XCD"
  GSH_LIB_PATH=${lib:+$ROOT/$lib} GSH_ACQ_LANES=$1 GSH_OC_WG_PER_XCD=$2 timeout 300 python profiles/ab/acq_ab.py 2>&1 | tail -1
done; done
} > $OUT/acq_bound.txt 2>&1
cat $OUT/acq_bound.txt
{
rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|mclk|fclk|socclk|Power" | head -8
echo "== running"
python profiles/ab/r05/acq_long.py 10000 5 > $OUT/acq_long.txt 2>&1 &
PID=$!
sleep 2.5
for i in 1 2 3; do rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|mclk|fclk|Power" | head -6; sleep 0.5; done
wait $PID
cat $OUT/acq_long.txt
} > $OUT/acq_clocks.txt 2>&1
cat $OUT/acq_clocks.txt
