#!/bin/bash
# the tree with two floors per instruction in the paired trip (GSH_MC_PKRTZ = 1, the default): the GPU suite (failures in full), then the bench line
cd /root/repo; mkdir -p gpurun_out/r06
{
timeout 1800 python -m pytest tests -x -q -m gpu > /tmp/suite.log 2>&1
grep -E "FAIL|passed|failed|^ERROR|Error" /tmp/suite.log | cut -c1-1500 | tail -15
timeout 1200 python bench.py > gpurun_out/r06/bench_s26.json 2> gpurun_out/r06/bench_s26.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06/bench_s26.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("metric", "value", "unit", "ms_per_step", "n_gpus")})
print("roofline", d.get("roofline"))
print("cpu_baseline", {k: d["cpu_baseline"][k] for k in ("value", "unit", "cores", "kind")})
for k in ("acquisition", "closed_loop", "closed_loop_cooperating", "valu_issue_frac", "hbm_unique_frac", "contract_hbm_rate_over_peak"):
    print(k, d.get(k) if not isinstance(d.get(k), dict) else {a: b for a, b in d[k].items() if not isinstance(b, (dict, list))})
PY
} > gpurun_out/r06/session26.txt 2>&1
cat gpurun_out/r06/session26.txt
