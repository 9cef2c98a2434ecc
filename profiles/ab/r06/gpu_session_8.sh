#!/bin/bash
# round 6, session 8: the library with both work-group sizes: tracking tests, A/B by GSH_MC_WG, configs 4 / 5, then the bench line
cd /root/repo
mkdir -p gpurun_out/r06
{
python -m pytest tests/test_tracking_gpu.py tests/test_tracking_configs_gpu.py tests/test_stream_group_gpu.py -m gpu -x -q 2>&1 | tail -3
GSH_MC_WG=128 python -m pytest tests/test_tracking_gpu.py tests/test_tracking_configs_gpu.py -m gpu -x -q 2>&1 | tail -3
for i in 1 2; do
GSH_MC_WG=256 python profiles/ab/mcorr_ab.py
python profiles/ab/mcorr_ab.py
done
python profiles/config_rates.py 2>&1 | grep "splits  0"
python bench.py --no-dropin --cpu-seconds 3 > gpurun_out/r06/bench_s8.json 2> gpurun_out/r06/bench_s8.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06/bench_s8.json"))
print(json.dumps(d["summary"]))
print(d["value"], d["roofline"]["kernel"], d["roofline"]["kernel_ms"])
PY
} > gpurun_out/r06/session8.txt 2>&1
grep -v amdgpu.ids gpurun_out/r06/session8.txt | tail -40
