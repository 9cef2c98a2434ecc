#!/bin/bash
# round 6, session 9: the latch test, the multi-rank tests, the whole bench line (with the drop-in leg and the CPU sweep)
cd /root/repo
mkdir -p gpurun_out/r06
{
python -m pytest tests/test_tracking_loop_gpu.py -m gpu -x -q -k "stamp_ahead or lock_detectors" 2>&1 | tail -3
python bench.py > gpurun_out/r06/bench_s9.json 2> gpurun_out/r06/bench_s9.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06/bench_s9.json"))
print(json.dumps(d["summary"], indent=0))
for k in ("traffic_source", "contract_hbm_rate_over_peak", "hbm_unique_frac", "valu_issue_frac", "hbm_read_probe_GBs", "hbm_read_probe_frac_of_nominal"):
    print(k, d.get(k))
print(json.dumps(d["cpu_baseline"], indent=0)[:1500])
PY
} > gpurun_out/r06/session9.txt 2>&1
grep -v amdgpu.ids gpurun_out/r06/session9.txt | tail -80
