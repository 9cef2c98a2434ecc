R=$PWD
for conf in lock sync; do for rep in 1 2; do
  GSH_LOOP_AB_CONF=$conf GSH_LIB_PATH=$R/build/variants/lib_prev.so python profiles/ab/closed_loop_ab.py 2>&1 | tail -2 | sed "s/^/previous: /"
  GSH_LOOP_AB_CONF=$conf python profiles/ab/closed_loop_ab.py 2>&1 | tail -2 | sed "s/^/current:  /"
done; done
