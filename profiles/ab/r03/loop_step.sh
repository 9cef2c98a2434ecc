# closed-loop kernel: current tree against the library of the previous commit (build/variants/lib_prev.so if present), phase breakdowns, loop tests
R=$PWD
for rep in 1 2; do
  [ -f build/variants/lib_prev.so ] && GSH_LIB_PATH=$R/build/variants/lib_prev.so python profiles/ab/closed_loop_ab.py 2>&1 | tail -2 | sed "s/^/previous: /"
  python profiles/ab/closed_loop_ab.py 2>&1 | tail -2 | sed "s/^/current:  /"
done
[ -f build/variants/lib_trkprof.so ] && GSH_PHASE_DETAIL=1 GSH_LIB_PATH=$R/build/variants/lib_trkprof.so python profiles/ab/closed_loop_phases.py 2>&1 | tail -2
[ -f build/variants/lib_trkprof2.so ] && GSH_PHASE_DETAIL=2 GSH_LIB_PATH=$R/build/variants/lib_trkprof2.so python profiles/ab/closed_loop_phases.py 2>&1 | tail -2
[ -f build/variants/lib_prev.so ] && GSH_LIB_PATH=$R/build/variants/lib_prev.so python profiles/ab/r03/loop_records.py /tmp/rec_prev.bin && python profiles/ab/r03/loop_records.py /tmp/rec_cur.bin && cmp /tmp/rec_prev.bin /tmp/rec_cur.bin && echo "records of both builds are byte-identical ($(stat -c %s /tmp/rec_cur.bin) bytes)"
timeout 900 python -m pytest tests/test_tracking_loop_gpu.py tests/test_symbol_sync.py tests/test_trk_dump.py tests/test_config1_file_input_gpu.py -m gpu -q -x 2>&1 | tail -3
