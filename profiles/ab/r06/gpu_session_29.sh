#!/bin/bash
# profiles of the round's final tree (rocprofv3 kernel trace + the counter passes of profiles/run_profiles_r06.sh), then the bench line
cd /root/repo; mkdir -p gpurun_out/r06
bash profiles/run_profiles_r06.sh r06 > gpurun_out/prof_r06.log 2>&1
cp profiles/r06_summary.txt profiles/pmc_r06.json gpurun_out/ 2>/dev/null
du -sh gpurun_out/prof_r06
# what travels back is bounded (64 MiB): the per-dispatch tables stay on the box, the per-kernel statistics and the logs come home
find gpurun_out/prof_r06 -name "*kernel_trace.csv" -size +1M -delete
find gpurun_out/prof_r06 -name "*counter_collection.csv" -delete
find gpurun_out/prof_r06 -name "*agent_info.csv" -delete
du -sh gpurun_out/prof_r06
tail -5 gpurun_out/prof_r06.log
timeout 1200 python bench.py > gpurun_out/r06/bench_final2.json 2> gpurun_out/r06/bench_final2.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06/bench_final2.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("metric", "value", "unit", "ms_per_step", "n_gpus")})
print("kernel_ms", d["roofline"]["kernel_ms"], "frac", d["roofline"]["frac"], "valu_issue_frac", d.get("valu_issue_frac"))
PY
timeout 600 python profiles/ab/r06/mcorr_two_streams.py 2>&1 | grep -v amdgpu > gpurun_out/r06/session30.txt
cat gpurun_out/r06/session30.txt
