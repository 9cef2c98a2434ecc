#!/bin/bash
# the round's final tree: the whole GPU suite twice (failures in full), the profiles (rocprofv3 kernel trace + counter passes), the bench line
cd /root/repo; mkdir -p gpurun_out/r06
{
for i in 1 2; do
  timeout 1500 python -m pytest tests -x -q -m gpu > /tmp/suite_$i.log 2>&1
  echo "suite run $i: $(grep -E ' passed| failed' /tmp/suite_$i.log | tail -1)"
  grep -E "^FAILED|^ERROR|FAIL " /tmp/suite_$i.log | cut -c1-1500 | head -12
done
cp /tmp/suite_2.log gpurun_out/r06/gpu_suite_final3.log
} > gpurun_out/r06/session37.txt 2>&1
bash profiles/run_profiles_r06.sh r06 > gpurun_out/prof_r06.log 2>&1
cp profiles/r06_summary.txt profiles/pmc_r06.json gpurun_out/ 2>/dev/null
find gpurun_out/prof_r06 -name "*kernel_trace.csv" -size +1M -delete
find gpurun_out/prof_r06 -name "*counter_collection.csv" -delete
find gpurun_out/prof_r06 -name "*agent_info.csv" -delete
timeout 1200 python bench.py > gpurun_out/r06/bench_final4.json 2> gpurun_out/r06/bench_final4.err
python - >> gpurun_out/r06/session37.txt <<'PY'
import json
d = json.loads(open("gpurun_out/r06/bench_final4.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("metric", "value", "unit", "ms_per_step", "n_gpus", "launches_in_flight", "value_single_stream")})
print("kernel_ms", d["roofline"]["kernel_ms"], "frac", d["roofline"]["frac"], "valu_issue_frac", d.get("valu_issue_frac"))
print(d["summary"])
PY
cat gpurun_out/r06/session37.txt
