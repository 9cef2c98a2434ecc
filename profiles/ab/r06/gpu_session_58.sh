#!/bin/bash
# profiles of the round's final tree (rocprofv3 kernel trace + the counter passes of profiles/run_profiles_r06.sh)
cd /root/repo; mkdir -p gpurun_out/r06
bash profiles/run_profiles_r06.sh r06 > gpurun_out/prof_r06.log 2>&1
cp profiles/r06_summary.txt profiles/pmc_r06.json gpurun_out/ 2>/dev/null
find gpurun_out/prof_r06 -name "*kernel_trace.csv" -size +1M -delete
find gpurun_out/prof_r06 -name "*counter_collection.csv" -delete
find gpurun_out/prof_r06 -name "*agent_info.csv" -delete
tail -5 gpurun_out/prof_r06.log > gpurun_out/r06/session58.txt
grep -n "subcell_dit\|combine_dit\|mcorr_kernel_t128\|oc_cell_kernel" profiles/r06_summary.txt | head -12 >> gpurun_out/r06/session58.txt
cat gpurun_out/r06/session58.txt
