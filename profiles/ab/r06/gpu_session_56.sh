#!/bin/bash
# 50 000 points: the persistent sub-cell launch had 55 work-groups per XCD x 6 sub-cells (two rounds on 32 compute units, the second 23 / 32 full); sub-cells per work-group
cd /root/repo; mkdir -p gpurun_out/r06
{
for rep in 1 2; do
for cfg in "6 28" "12 28" "11 32" "10 32" "14 24" "24 14" "21 16"; do set -- $cfg
  echo "== sub-cells per work-group $1, work-groups per XCD at least $2"; GSH_OC_CELLS_PER_WG=$1 GSH_OC_WG_PER_XCD=$2 python profiles/ab/r06/acq_128k.py 50000:50e6 2>&1 | grep "^N ="
done; done
} > gpurun_out/r06/session56.txt 2>&1
cat gpurun_out/r06/session56.txt
