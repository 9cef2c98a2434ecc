# closed-loop kernel, round 4: layouts of the serial section (build/variants/lib_<tag>.so = -DGSH_TRK_SERIAL_WAVES / -DGSH_TRK_PREFIX_ALL), against the previous commit's library
R=$PWD
for rep in 1 2; do for conf in lock ""; do
  for lib in "$@"; do
    if [ "$lib" = current ]; then GSH_LOOP_AB_CONF=$conf python profiles/ab/closed_loop_ab.py 2>&1 | tail -2 | sed "s/^/current /"
    else GSH_LOOP_AB_CONF=$conf GSH_LIB_PATH=$R/build/variants/lib_$lib.so python profiles/ab/closed_loop_ab.py 2>&1 | tail -2 | sed "s/^/$lib /; s#/root/repo/build/variants/##"; fi
  done
done; done
