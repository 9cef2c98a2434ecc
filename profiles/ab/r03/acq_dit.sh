# split plans: decimation in time (default for S >= 4) against decimation in frequency (GSH_OC_DIT_MIN_S=0) and DIT for every split plan (=2)
for v in 0 4 2; do
  echo "== GSH_OC_DIT_MIN_S=$v"
  GSH_OC_DIT_MIN_S=$v timeout 300 python profiles/acq_ab.py 32000 50000 64000 80000 100000 128000 2>&1 | grep -v amdgpu.ids
done
timeout 900 python -m pytest tests/test_acquisition_gpu.py tests/test_pcps_detectors_gpu.py -m gpu -x -q 2>&1 | tail -4
GSH_OC_DIT_MIN_S=2 timeout 900 python -m pytest tests/test_acquisition_gpu.py -m gpu -x -q 2>&1 | tail -3
