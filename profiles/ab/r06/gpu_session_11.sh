#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r06
{
python -m pytest tests/test_acquisition_gpu.py -m gpu -x -q -k "split or 128000 or long or dit" 2>&1 | tail -2
for w in 28 32; do for c in 6 30; do
echo "== persistent sub-cells, no touch, $w work-groups per XCD, $c cells per work-group"; GSH_OC_WG_PER_XCD=$w GSH_OC_CELLS_PER_WG=$c python profiles/ab/r06/acq_split_scale.py 2>&1 | grep "128000, 32 PRN"
done; done
echo "== no operand prefetch"; GSH_OC_PREFETCH=0 python profiles/ab/r06/acq_split_scale.py 2>&1 | grep "128000, 32 PRN"
echo "== one work-group per sub-cell"; GSH_OC_DIT_PERSIST=0 python profiles/ab/r06/acq_split_scale.py 2>&1 | grep "128000, 32 PRN"
} > gpurun_out/r06/session11.txt 2>&1
grep -v amdgpu.ids gpurun_out/r06/session11.txt | tail -40
