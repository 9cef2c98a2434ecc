#!/bin/bash
# Round-5 measurements, part 30: exchange 2 writing a phase before it reads the previous one (-DGSH_OC_EX2_WRITE_FIRST) against reading first (shipped); work-groups per XCD once more
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05p; mkdir -p $OUT
{
for rep in 1 2 3; do
  echo "== shipped";      timeout 300 python profiles/ab/acq_ab.py 2>&1 | tail -1
  echo "== write first";  GSH_LIB_PATH=$ROOT/build/variants/lib_oc_wfirst.so timeout 300 python profiles/ab/acq_ab.py 2>&1 | tail -1
done
for w in 30 32; do echo "== shipped, $w work-groups per XCD"; GSH_OC_WG_PER_XCD=$w timeout 300 python profiles/ab/acq_ab.py 2>&1 | tail -1; done
} > $OUT/acq_ex2_order.txt 2>&1
cat $OUT/acq_ex2_order.txt
