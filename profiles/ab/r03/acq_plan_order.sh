# 25 000 points: the three orderings of the radices 25, 25, 40 (build/variants/lib_plan*.so: -D'GSH_OC_PLANS(X)=X(25,40,25)' etc.)
R=$PWD
for rep in 1 2; do
  python profiles/acq_ab.py 25000 2>&1 | grep -v amdgpu
  GSH_LIB_PATH=$R/build/variants/lib_plan254025.so python profiles/acq_ab.py 25000 2>&1 | grep -v amdgpu
  GSH_LIB_PATH=$R/build/variants/lib_plan402525.so python profiles/acq_ab.py 25000 2>&1 | grep -v amdgpu
done
