#!/bin/bash
# round 6, session 1: where the batched correlator stands -- fixed cost per job (T = a + b n) and 128- / 64-thread work-groups of the same source
cd /root/repo; export TMPDIR=/tmp
mkdir -p gpurun_out/r06
{
python profiles/ab/mcorr_fixed_cost.py
for i in 1 2; do
python profiles/ab/mcorr_ab.py
GSH_LIB_PATH=build/variants/lib_t128.so python profiles/ab/mcorr_ab.py
GSH_LIB_PATH=build/variants/lib_t64.so python profiles/ab/mcorr_ab.py
done
} > gpurun_out/r06/session1.txt 2>&1
tail -30 gpurun_out/r06/session1.txt
