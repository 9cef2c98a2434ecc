#!/bin/bash
# bench.py with two tracking launches in flight (the default from here on): the whole line, then the bench's own tests
cd /root/repo; mkdir -p gpurun_out/r06
{
timeout 1200 python bench.py > gpurun_out/r06/bench_final3.json 2> gpurun_out/r06/bench_final3.err
tail -3 gpurun_out/r06/bench_final3.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06/bench_final3.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("metric", "value", "unit", "ms_per_step", "n_gpus", "launches_in_flight", "value_single_stream", "ms_per_step_single_stream")})
print("kernel_ms", d["roofline"]["kernel_ms"], "frac", d["roofline"]["frac"], "spot", d["spot_check"])
print(d["summary"])
PY
timeout 900 python -m pytest tests/test_bench_stdout.py tests/test_stream_group_multi_gpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed|^FAILED|^ERROR" | tail -3
} > gpurun_out/r06/session31.txt 2>&1
cat gpurun_out/r06/session31.txt
