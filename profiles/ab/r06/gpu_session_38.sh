#!/bin/bash
# the batched kernels issue their first loads ahead of the seed factors and the paired-tap judgement (GSH_MC_EARLY_LOADS_BANK = 1, build/variants/lib_el.so) against the shipped order
cd /root/repo; mkdir -p gpurun_out/r06
{
GSH_LIB_PATH=/root/repo/build/variants/lib_el.so timeout 900 python -m pytest tests/test_tracking_gpu.py tests/test_tracking_configs_gpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed|^FAILED|^ERROR|Error" | tail -5
for tag in shipped el shipped el shipped el; do
  if [ $tag = shipped ]; then unset GSH_LIB_PATH; else export GSH_LIB_PATH=/root/repo/build/variants/lib_$tag.so; fi
  timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-other-configs --no-dropin --no-acq > /tmp/b.json 2>/tmp/b.err
  python - $tag <<'PY'
import json, sys
d = json.loads(open("/tmp/b.json").read().strip().splitlines()[-1])
print("== %-8s %.1f M correlators/s two in flight, %.1f M one at a time, kernel %.1f us, frac %.3f" % (sys.argv[1], d["value"] / 1e6, d["value_single_stream"] / 1e6, d["roofline"]["kernel_ms"] * 1e3, d["roofline"]["frac"]))
PY
done
} > gpurun_out/r06/session38.txt 2>&1
cat gpurun_out/r06/session38.txt
