mkdir -p gpurun_out/r02c
./tests/host/test_tracking_adapters 2>&1 | grep -v "^Tracking of\|histogram bit\|secondary code locked" | tail -30 | tee gpurun_out/r02c/tracking_adapters.log
python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee gpurun_out/r02c/gpu_suite.log
