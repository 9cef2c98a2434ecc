# chunked decimation in time (two chunks in flight, scratch in the Infinity Cache): every split plan (GSH_OC_DIT_MIN_S=2), the default (4), and DIF (0)
for v in 2 4 0; do
  echo "== GSH_OC_DIT_MIN_S=$v"
  GSH_OC_DIT_MIN_S=$v timeout 300 python profiles/acq_ab.py 32000 50000 64000 100000 128000 2>&1 | grep -v amdgpu.ids
done
for mb in 48 160; do echo "== chunk budget $mb MB, min_s 2"; GSH_OC_DIT_MIN_S=2 GSH_OC_DIT_CHUNK_MB=$mb timeout 300 python profiles/acq_ab.py 50000 128000 2>&1 | grep -v amdgpu.ids; done
GSH_OC_DIT_MIN_S=2 timeout 900 python -m pytest tests/test_acquisition_gpu.py tests/test_pcps_detectors_gpu.py -m gpu -x -q 2>&1 | tail -3
timeout 900 python -m pytest tests/test_acquisition_gpu.py -m gpu -x -q 2>&1 | tail -3
