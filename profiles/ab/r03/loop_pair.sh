# closed-loop kernel with early-from-late trips (GSH_TRK_PAIRED_TAPS) against the shipped one, alternating on one box
R=$PWD
for rep in 1 2; do
  python profiles/ab/closed_loop_ab.py 2>&1 | tail -2 | sed "s/^/shipped: /"
  GSH_LIB_PATH=$R/build/variants/lib_trkpair.so python profiles/ab/closed_loop_ab.py 2>&1 | tail -2 | sed "s/^/paired:  /"
done
GSH_PHASE_DETAIL=2 GSH_LIB_PATH=$R/build/variants/lib_trkprof2.so python profiles/ab/closed_loop_phases.py 2>&1 | tail -2 | sed "s/^/shipped: /"
GSH_PHASE_DETAIL=2 GSH_LIB_PATH=$R/build/variants/lib_trkpairprof.so python profiles/ab/closed_loop_phases.py 2>&1 | tail -2 | sed "s/^/paired:  /"
GSH_LIB_PATH=$R/build/variants/lib_trkpair.so timeout 600 python -m pytest tests/test_tracking_loop_gpu.py tests/test_symbol_sync.py -m gpu -q -x 2>&1 | tail -3
