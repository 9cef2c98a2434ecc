R=$PWD
for rep in 1 2; do for f in c_base c_ilp c_mem; do GSH_LIB_PATH=$R/build/variants/lib_$f.so python profiles/acq_ab.py 25000 2>&1 | grep -v amdgpu; done; done
