# closed-loop kernel, round 4: the tree's library against build/variants/lib_prev.so (the library of the previous commit) -- lock detectors on / + symbol sync / off,
# the records of both byte for byte (profiles/ab/r03/loop_records.py), then the loop's GPU tests
R=$PWD
for conf in lock sync ""; do for rep in 1 2; do
  GSH_LOOP_AB_CONF=$conf GSH_LIB_PATH=$R/build/variants/lib_prev.so python profiles/ab/closed_loop_ab.py 2>&1 | tail -2 | sed "s/^/previous: /"
  GSH_LOOP_AB_CONF=$conf python profiles/ab/closed_loop_ab.py 2>&1 | tail -2 | sed "s/^/current:  /"
done; done
GSH_LIB_PATH=$R/build/variants/lib_prev.so python profiles/ab/r03/loop_records.py /tmp/rec_prev.bin && python profiles/ab/r03/loop_records.py /tmp/rec_cur.bin && cmp /tmp/rec_prev.bin /tmp/rec_cur.bin && echo "records of both builds are byte-identical ($(stat -c %s /tmp/rec_cur.bin) bytes)"
timeout 900 python -m pytest tests/test_tracking_loop_gpu.py tests/test_symbol_sync.py tests/test_trk_dump.py tests/test_tracking_live_gpu.py tests/test_config1_file_input_gpu.py -m gpu -q -x 2>&1 | tail -3
