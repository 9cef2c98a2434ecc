#!/bin/bash
# A/B on one box: the packed trip loop before (build/variants/lib_old.so: git HEAD~ of the change) and after -- edge clamps a real branch, no register copies in
# the prefetch queue, DPP wave sums, cheap margin wrap; closed loop: the same + one barrier fewer per correlation
cd "$(dirname "$0")/../.."
R=$(pwd)
for rep in 1 2 3; do
  GSH_LIB_PATH=$R/build/variants/lib_old.so python profiles/ab/mcorr_ab.py 2>&1 | tail -1 | sed "s/^/before: /"
  python profiles/ab/mcorr_ab.py 2>&1 | tail -1 | sed "s/^/after:  /"
done
GSH_LIB_PATH=$R/build/variants/lib_old.so python profiles/ab/closed_loop_ab.py 2>&1 | tail -2 | sed "s/^/before: /"
python profiles/ab/closed_loop_ab.py 2>&1 | tail -2 | sed "s/^/after:  /"
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -3
