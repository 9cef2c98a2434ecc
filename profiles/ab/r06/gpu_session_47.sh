#!/bin/bash
# compile-time non-temporal Z (S < 8), ordered lanes there: split shapes, acquisition tests, whole suite, bench line
cd /root/repo
mkdir -p gpurun_out/r06
{
for rep in 1 2; do python profiles/ab/r06/acq_128k.py 50000:50e6 100000:25e6 128000:32e6 200000:50e6 2>&1 | grep "^N ="; done
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -5
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -i "smoke"
python bench.py > gpurun_out/r06/bench47.json 2> gpurun_out/r06/bench47.err; echo "bench rc $?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06/bench47.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "n_gpus")}, d["roofline"]["kernel_ms"], d["roofline"]["frac"])
print(json.dumps(d["acquisition"]["split_plan_128000"]), json.dumps(d["acquisition"]["split_plan_50000"]))
print(json.dumps(d["summary"]))
PY
} > gpurun_out/r06/session47.txt 2>&1
cat gpurun_out/r06/session47.txt
