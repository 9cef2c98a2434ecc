#!/bin/bash
# round 6, session 6: work-group size of the batched kernel (256 / 128 / 64 threads per job; 64 with four paired-tap masks): time, then counters
cd /root/repo
mkdir -p gpurun_out/r06
{
for i in 1 2 3; do
python profiles/ab/mcorr_ab.py
GSH_LIB_PATH=build/variants/lib_t128.so python profiles/ab/mcorr_ab.py
GSH_LIB_PATH=build/variants/lib_t64.so python profiles/ab/mcorr_ab.py
done
bash profiles/ab/r06/pmc_mcorr.sh t256 python /root/repo/profiles/ab/mcorr_ab.py
bash profiles/ab/r06/pmc_mcorr.sh t128 GSH_LIB_PATH=/root/repo/build/variants/lib_t128.so python /root/repo/profiles/ab/mcorr_ab.py
bash profiles/ab/r06/pmc_mcorr.sh t64 GSH_LIB_PATH=/root/repo/build/variants/lib_t64.so python /root/repo/profiles/ab/mcorr_ab.py
GSH_LIB_PATH=build/variants/lib_t128.so python -m pytest tests/test_tracking_gpu.py tests/test_tracking_configs_gpu.py -m gpu -x -q 2>&1 | tail -3
GSH_LIB_PATH=build/variants/lib_t64.so python -m pytest tests/test_tracking_gpu.py tests/test_tracking_configs_gpu.py -m gpu -x -q 2>&1 | tail -3
} > gpurun_out/r06/session6.txt 2>&1
grep -v amdgpu.ids gpurun_out/r06/session6.txt | tail -40
rm -rf gpurun_out/r06/pmc_*
