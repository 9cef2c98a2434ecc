# dropin: 32 tracking blocks behind their adapters on one 25 Msps stream, 2 400 periods, side by side with the reference blocks (tests/host/test_tracking_adapters bench ...)
for k in 10 20 40; do ./tests/host/test_tracking_adapters bench 32 25000000 2400 $k 2>&1 | grep DROPIN_JSON | cut -c1-420; done
./tests/host/test_tracking_adapters bench 32 25000000 2400 20 2>&1 | grep DROPIN_JSON | cut -c1-420
./tests/host/test_tracking_adapters bench 32 4000000 6000 20 2>&1 | grep DROPIN_JSON | cut -c1-420
timeout 900 python -m pytest tests/test_tracking_adapters.py tests/test_tracking_loop_gpu.py tests/test_config1_file_input_gpu.py tests/test_sample_stream_gpu.py -m gpu -x -q 2>&1 | tail -3
