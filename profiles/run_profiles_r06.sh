#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repo root:  bash profiles/run_profiles_r06.sh <tag>
# rocprofv3 passes over the SAME bench.py command (shortened: fewer steps, no CPU leg), kernel-trace only in every --pmc pass, as
# /opt/skills/guides/MI355X_MICROARCH.md prescribes:
#   trace/            --kernel-trace --stats                       per-kernel durations
#   fetch/ write/     --pmc FETCH_SIZE ; --pmc WRITE_SIZE          HBM-side bytes (separate passes), calibrated with calib_*/ (1 GiB torch reduction / copy)
#   sq1/ sq2/ tcc/    SQ issue counters ; L2 (TCC) requests
# profiles/summarize_r02.py turns them into profiles/<tag>_summary.txt and profiles/pmc_r06.json (what bench.py reads).
# Round 3 adds: acq_shared/ acq_alone/  kernel traces of eight acquisition channels on one stream through the shared runtime / on their own handles
#               (tests/host/test_adapters acq_shared | acq_alone): the forward-transform launches of either, appended to the summary.
set -u
TAG=${1:-r06}
export GSH_BENCH_ACQ_SINGLE_STREAM=1   # one acquisition batch at a time: the trace then shows each kernel alone (bench.py)
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 2 --warmup 1 --settle-steps 1 --blocks-per-step 16 --launches-in-flight 1 --no-cpu-baseline --no-other-configs --no-dropin"   # one tracking launch at a time too: the trace shows each kernel alone, as roofline.kernel_ms is measured
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- $BENCH > $OUT/trace.json 2> $OUT/trace.err
for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE" \
            "sq1:SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
            "sq2:SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" \
            "tcc:TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr"; do
  name=${pass%%:*}; ctr=${pass#*:}
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/$name -o bench -- $BENCH > $OUT/$name.json 2> $OUT/$name.err
done
cat > /tmp/calib.py <<'PY'
import torch
x = torch.ones(1 << 28, dtype=torch.float32, device="cuda")   # 1 GiB, larger than the 256 MiB Infinity Cache
torch.cuda.synchronize()
for _ in range(3):
    s = x.sum()
y = torch.empty_like(x)
for _ in range(3):
    y.copy_(x)
torch.cuda.synchronize()
print(float(s))
PY
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/calib_fetch -o calib -- python /tmp/calib.py > /dev/null 2> $OUT/calib_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/calib_write -o calib -- python /tmp/calib.py > /dev/null 2> $OUT/calib_write.err
find $OUT -name "*.db" -delete
du -sh $OUT
for mode in acq_shared acq_alone; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$mode -o acq -- $ROOT/tests/host/test_adapters $mode > $OUT/$mode.log 2> $OUT/$mode.err
done
find $OUT -name "*.db" -delete
cd $ROOT && GSH_PMC_JSON=pmc_r06.json python profiles/summarize_r02.py $OUT $TAG
tail -12 $ROOT/profiles/${TAG}_summary.txt
