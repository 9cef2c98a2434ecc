R=$(pwd)
rocprofv3-avail list > gpurun_out/counters_avail_raw.txt 2>&1; grep -c . gpurun_out/counters_avail_raw.txt
GSH_MC_PACKED_BODY=1 bash profiles/pmc_kernel2.sh mcorr_kernel mc_pk1 python $R/profiles/ab/mcorr_ab.py | tail -70
GSH_MC_PACKED_BODY=0 bash profiles/pmc_kernel2.sh mcorr_kernel mc_pk0 python $R/profiles/ab/mcorr_ab.py | tail -70
