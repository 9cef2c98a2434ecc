#!/bin/bash
# (a) issue rates of the candidates for the trip's floor() (profiles/ubench/valu_rate); (b) the channel test of the GPU suite through pytest, five times, failures in full
cd /root/repo; mkdir -p gpurun_out/r06
{
./profiles/ubench/valu_rate | grep -E "waves/SIMD=6"
for i in 1 2 3 4 5; do
  timeout 900 python -m pytest tests/test_tracking_adapters.py -x -q -m gpu -k "channel_life_and_churn" 2>&1 | grep -E "FAIL|passed|failed|churn:" | cut -c1-1500 | tail -12
done
} > gpurun_out/r06/session24.txt 2>&1
cat gpurun_out/r06/session24.txt
