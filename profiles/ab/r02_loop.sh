R=$(pwd); mkdir -p gpurun_out/r02e
python -m pytest tests -m gpu -x -q 2>&1 | tail -2
(for f in build/variants/lib_x*.so; do echo $f; GSH_LIB_PATH=$R/$f python profiles/ab/closed_loop_phases.py 2>&1 | tail -1; done; python profiles/ab/closed_loop_ab.py 2>&1 | tail -2;  python profiles/ab/mcorr_ab.py 2>&1 | tail -1) | tee gpurun_out/r02e/loop_phase_ab2.log
