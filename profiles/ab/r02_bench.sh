mkdir -p gpurun_out/r02d
./tests/host/test_tracking_adapters 2>&1 | grep -v "^Tracking of\|histogram bit\|secondary code locked\|^GPS L1 C/A tracking" | tail -12 | tee gpurun_out/r02d/tracking_adapters.log
./tests/host/test_adapters 2>&1 | tail -12 | tee gpurun_out/r02d/adapters.log
python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee gpurun_out/r02d/gpu_suite.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r02d/bench.json 2> gpurun_out/r02d/bench.err; tail -c 600 gpurun_out/r02d/bench.err; python - <<'PY'
import json
b=json.loads(open('gpurun_out/r02d/bench.json').read().strip().splitlines()[-1])
print({k:b[k] for k in ('value','ms_per_step','steps')}, b['config']['timed_region_s'])
print('roofline', {k:v for k,v in b['roofline'].items() if k not in ('pmc',)})
print('cpu', b.get('cpu_baseline'))
print('pcie', b.get('pcie_inclusive'))
print('acq', {k:v for k,v in b.get('acquisition',{}).items() if k!='cpu_baseline'})
print('closed', b.get('closed_loop'))
PY
GSH_BENCH_FORCE_DIST=1 python bench.py --steps 3 --warmup 1 --blocks-per-step 32 --no-cpu-baseline --no-acq 2>&1 | tail -c 1500
