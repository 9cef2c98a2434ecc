timeout 300 python profiles/acq_ab.py 64000 80000 100000 128000 200000 2>&1 | grep -v amdgpu.ids
timeout 900 python -m pytest tests/test_acquisition_gpu.py tests/test_pcps_detectors_gpu.py tests/test_fir_filter_gpu.py -m gpu -x -q 2>&1 | tail -3
