#!/bin/bash
# round 6, session 5: vector instructions per wave against the window length: VALU / wave = F + P * trips
cd /root/repo
mkdir -p gpurun_out/r06
{
for n in 1024 2048 4096 12288 25000; do
bash profiles/ab/r06/pmc_mcorr.sh r05_n$n GSH_LIB_PATH=/root/repo/build/variants/lib_r05.so python /root/repo/profiles/ab/r06/mcorr_n.py $n
bash profiles/ab/r06/pmc_mcorr.sh c1_n$n python /root/repo/profiles/ab/r06/mcorr_n.py $n 1
bash profiles/ab/r06/pmc_mcorr.sh c10_n$n python /root/repo/profiles/ab/r06/mcorr_n.py $n 10
done
} > gpurun_out/r06/session5.txt 2>&1
grep -v amdgpu.ids gpurun_out/r06/session5.txt | grep -o "^[a-z0-9_]* \|VALU/wave.*\|kernel avg ns [0-9.]*" | paste - - - | tail -40
rm -rf gpurun_out/r06/pmc_*
