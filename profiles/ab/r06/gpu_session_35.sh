#!/bin/bash
# the plain trips of the non-paired loops (five-tap / pilot + data / windowed batched kernels, the closed-loop kernel) as a counted range with the one-comparison next-load
# test (the shipped library) against GSH_MC_RUNLEN = 0 (build/variants/lib_norun.so): bit-exactness and closed-loop tests, then closed loop and configs 4 / 5, alternating
cd /root/repo; mkdir -p gpurun_out/r06
{
timeout 1500 python -m pytest tests/test_tracking_gpu.py tests/test_tracking_configs_gpu.py tests/test_tracking_loop_gpu.py tests/test_symbol_sync.py tests/test_trk_dump.py -x -q -m gpu > /tmp/t.log 2>&1
grep -E "FAIL|passed|failed|^ERROR|Error" /tmp/t.log | cut -c1-800 | tail -6
for tag in norun shipped norun shipped norun shipped; do
  if [ $tag = shipped ]; then unset GSH_LIB_PATH; else export GSH_LIB_PATH=/root/repo/build/variants/lib_$tag.so; fi
  echo "== $tag"; timeout 300 python profiles/ab/r06/closed_loop_kc.py 2>&1 | grep -v amdgpu | tail -2
done
for tag in norun shipped norun shipped; do
  if [ $tag = shipped ]; then unset GSH_LIB_PATH; else export GSH_LIB_PATH=/root/repo/build/variants/lib_$tag.so; fi
  timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-dropin > /tmp/b.json 2>/tmp/b.err
  python - $tag <<'PY'
import json, sys
d = json.loads(open("/tmp/b.json").read().strip().splitlines()[-1])
s = d["summary"]; oc = d.get("other_configs", {})
print("== %-8s tracking %.1f / %.1f M; closed loop %.3f us (detectors %.3f, live %.3f, 256 ch %.3f), config 4 closed %.2f us (4 wg %.2f); config 4 batched %.4f ms, config 5 share %.4f ms" % (
    sys.argv[1], s["tracking_Mcorr_s"], s["tracking_Mcorr_s_single_stream"], s["closed_loop_us"], s["closed_loop_detectors_us"], s["closed_loop_live_us"], s["closed_loop_256ch_us"],
    s["closed_loop_config4_us"], s["closed_loop_config4_4wg_us"], oc["config4_galileo_e1_50ch_32Msps"]["ms_per_launch"], oc["config5_share_32_of_256ch_50Msps"]["ms_per_launch"]))
PY
done
} > gpurun_out/r06/session35.txt 2>&1
cat gpurun_out/r06/session35.txt
