#!/bin/bash
# at most 32 persistent work-groups per XCD: split shapes, a wide single-plan grid (32 PRNs x 81 bins at 25 000 points), acquisition tests, whole suite, bench line
cd /root/repo; mkdir -p gpurun_out/r06
{
for rep in 1 2; do python profiles/ab/r06/acq_128k.py 50000:50e6 100000:25e6 128000:32e6 2>&1 | grep "^N ="; done
python -m pytest tests -m gpu -x -q > /tmp/suite.log 2>&1; grep -E " passed| failed" /tmp/suite.log | tail -1; grep -E "^FAIL|^E  " /tmp/suite.log | cut -c1-1500 | head -20
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -i "smoke"
timeout 1200 python bench.py > gpurun_out/r06/bench57.json 2> gpurun_out/r06/bench57.err; echo "bench rc $?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06/bench57.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("metric", "value", "unit", "ms_per_step", "n_gpus")})
print("kernel_ms", d["roofline"]["kernel_ms"], "frac", d["roofline"]["frac"], "valu_issue_frac", d.get("valu_issue_frac"))
print(json.dumps(d["acquisition"]["split_plan_128000"])); print(json.dumps(d["acquisition"]["split_plan_50000"]))
print(json.dumps(d["summary"]))
PY
} > gpurun_out/r06/session57.txt 2>&1
cut -c1-1600 gpurun_out/r06/session57.txt
