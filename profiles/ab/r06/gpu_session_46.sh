#!/bin/bash
# the tree with: combine launch without hand-off, non-temporal Z on S < 8, ordered lanes -- split shapes, whole GPU suite, smoke, bench line
cd /root/repo
mkdir -p gpurun_out/r06
{
python profiles/ab/r06/acq_128k.py 50000:50e6 100000:25e6 128000:32e6 200000:50e6 2>&1 | grep "^N ="
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -5
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -i "smoke"
python bench.py > gpurun_out/r06/bench46.json 2> gpurun_out/r06/bench46.err; echo "bench rc $?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06/bench46.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "n_gpus")}, d["roofline"]["kernel_ms"], d["roofline"]["frac"])
for k, v in d.items():
    if isinstance(v, dict) and k not in ("config", "roofline"):
        print(k, json.dumps(v)[:600])
PY
} > gpurun_out/r06/session46.txt 2>&1
cat gpurun_out/r06/session46.txt
