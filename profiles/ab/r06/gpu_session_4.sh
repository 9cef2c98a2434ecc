#!/bin/bash
# round 6, session 4: instruction counts per wave -- round-5 kernel, chain kernel with chains of 1 / 10, factor tables off
cd /root/repo
mkdir -p gpurun_out/r06
{
bash profiles/ab/r06/pmc_mcorr.sh r05 GSH_LIB_PATH=/root/repo/build/variants/lib_r05.so python /root/repo/profiles/ab/mcorr_ab.py
bash profiles/ab/r06/pmc_mcorr.sh chain1 GSH_MC_CHAIN=1 python /root/repo/profiles/ab/mcorr_ab.py
bash profiles/ab/r06/pmc_mcorr.sh chain1_nofac GSH_MC_CHAIN=1 GSH_MC_FAC=0 python /root/repo/profiles/ab/mcorr_ab.py
bash profiles/ab/r06/pmc_mcorr.sh chain10 GSH_MC_CHAIN=10 python /root/repo/profiles/ab/mcorr_ab.py
bash profiles/ab/r06/pmc_mcorr.sh chain5 GSH_MC_CHAIN=5 python /root/repo/profiles/ab/mcorr_ab.py
} > gpurun_out/r06/session4.txt 2>&1
grep -v amdgpu.ids gpurun_out/r06/session4.txt | tail -40
