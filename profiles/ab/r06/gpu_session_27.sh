#!/bin/bash
# same-box A/B of the bench's own kernel figure: the paired trip with v_cvt_flr_i32_f32 (build/variants/lib_flr.so = the round's earlier trees) against two floors per
# instruction (the shipped library), three alternations of a shortened bench.py
cd /root/repo; mkdir -p gpurun_out/r06
{
for tag in flr shipped flr shipped flr shipped; do
  if [ $tag = shipped ]; then unset GSH_LIB_PATH; else export GSH_LIB_PATH=/root/repo/build/variants/lib_$tag.so; fi
  timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-other-configs --no-dropin --no-acq > /tmp/b.json 2>/tmp/b.err
  python - $tag <<'PY'
import json, sys
d = json.loads(open("/tmp/b.json").read().strip().splitlines()[-1])
print("== %-8s %.1f M correlators/s, kernel %.1f us, frac %.3f" % (sys.argv[1], d["value"] / 1e6, d["roofline"]["kernel_ms"] * 1e3, d["roofline"]["frac"]))
PY
done
} > gpurun_out/r06/session27.txt 2>&1
cat gpurun_out/r06/session27.txt
