# correlator kernel under different instruction-scheduling strategies of the compiler (build/variants/lib_f_*.so: multicorrelator.hip recompiled with the flag)
R=$PWD
for rep in 1 2; do
  python profiles/ab/mcorr_ab.py 2>&1 | tail -1
  for f in build/variants/lib_f_*.so; do GSH_LIB_PATH=$R/$f python profiles/ab/mcorr_ab.py 2>&1 | tail -1; done
done
