# phase clocks of the closed-loop kernel: build/variants/lib_<tag>prof.so (-DGSH_TRK_PROFILE=1: the serial section) and lib_<tag>prof2.so (=2: the correlation)
R=$PWD
for conf in "" lock; do for tag in "$@"; do
  GSH_LOOP_AB_CONF=$conf GSH_PHASE_DETAIL=1 GSH_LIB_PATH=$R/build/variants/lib_${tag}prof.so python profiles/ab/closed_loop_phases.py 2>&1 | tail -2
  GSH_LOOP_AB_CONF=$conf GSH_PHASE_DETAIL=2 GSH_LIB_PATH=$R/build/variants/lib_${tag}prof2.so python profiles/ab/closed_loop_phases.py 2>&1 | tail -2
done; done
