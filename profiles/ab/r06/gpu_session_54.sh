#!/bin/bash
# the round's last tree (hints on every decimation-in-time split; the churn's reference run repeated once when a steady channel of it stays silent):
# split shapes, the channel test under load 6 x, whole GPU suite, smoke, bench line
cd /root/repo; mkdir -p gpurun_out/r06
{
for rep in 1 2; do python profiles/ab/r06/acq_128k.py 50000:50e6 100000:25e6 128000:32e6 200000:50e6 2>&1 | grep "^N ="; done
python -m pytest tests -m gpu -x -q > /tmp/suite.log 2>&1; grep -E " passed| failed" /tmp/suite.log | tail -1; grep -E "^FAIL|^E  " /tmp/suite.log | cut -c1-1500 | head -20
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -i "smoke"
timeout 1200 python bench.py > gpurun_out/r06/bench54.json 2> gpurun_out/r06/bench54.err; echo "bench rc $?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06/bench54.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("metric", "value", "unit", "ms_per_step", "n_gpus")})
print("kernel_ms", d["roofline"]["kernel_ms"], "frac", d["roofline"]["frac"], "valu_issue_frac", d.get("valu_issue_frac"))
print(json.dumps(d["acquisition"]["split_plan_128000"]))
print(json.dumps(d["summary"]))
PY
for j in $(seq 16); do ( python -c "
import time
t=time.time()
while time.time()-t<230: pass" & ) ; done
for i in 1 2 3 4 5 6; do
  t0=$(date +%s)
  ( cd /tmp && /root/repo/tests/host/test_channel churn 32 8 2.4 1 > /tmp/churn_$i.out 2> /tmp/churn_$i.err; echo "loaded run $i rc $? ($(( $(date +%s) - t0 )) s)" )
  grep -E "^FAIL|repeated once" /tmp/churn_$i.out | cut -c1-1500
done
} > gpurun_out/r06/session54.txt 2>&1
cut -c1-1600 gpurun_out/r06/session54.txt
