#!/bin/bash
# the round's last tree: split shapes, whole GPU suite, smoke, the profile passes (r06_summary.txt, pmc_r06.json), the bench line
cd /root/repo; mkdir -p gpurun_out/r06
{
for rep in 1 2; do python profiles/ab/r06/acq_128k.py 50000:50e6 100000:25e6 128000:32e6 200000:50e6 2>&1 | grep "^N ="; done
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -5
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -i "smoke"
} > gpurun_out/r06/session50.txt 2>&1
bash profiles/run_profiles_r06.sh r06 > gpurun_out/prof_r06.log 2>&1
cp profiles/r06_summary.txt profiles/pmc_r06.json gpurun_out/ 2>/dev/null
find gpurun_out/prof_r06 -name "*kernel_trace.csv" -size +1M -delete
find gpurun_out/prof_r06 -name "*counter_collection.csv" -delete
find gpurun_out/prof_r06 -name "*agent_info.csv" -delete
tail -5 gpurun_out/prof_r06.log >> gpurun_out/r06/session50.txt
timeout 1200 python bench.py > gpurun_out/r06/bench50.json 2> gpurun_out/r06/bench50.err; echo "bench rc $?" >> gpurun_out/r06/session50.txt
python - >> gpurun_out/r06/session50.txt <<'PY'
import json
d = json.loads(open("gpurun_out/r06/bench50.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("metric", "value", "unit", "ms_per_step", "n_gpus")})
print("kernel_ms", d["roofline"]["kernel_ms"], "frac", d["roofline"]["frac"], "valu_issue_frac", d.get("valu_issue_frac"))
print(json.dumps(d["acquisition"]["split_plan_128000"]))
print(json.dumps(d["summary"]))
PY
cat gpurun_out/r06/session50.txt
