R=$PWD
for rep in 1 2 3; do
  GSH_LIB_PATH=$R/build/variants/lib_prev.so python profiles/ab/closed_loop_ab.py 2>&1 | tail -2 | sed "s/^/previous: /"
  python profiles/ab/closed_loop_ab.py 2>&1 | tail -2 | sed "s/^/current:  /"
done
GSH_PHASE_DETAIL=2 GSH_LIB_PATH=$R/build/variants/lib_trkprof2.so python profiles/ab/closed_loop_phases.py 2>&1 | tail -2
