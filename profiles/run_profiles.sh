#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repo root:  bash profiles/run_profiles.sh <tag>
# Produces under gpurun_out/prof_<tag>/ :
#   trace/   rocprofv3 --kernel-trace --stats of the default bench.py run (per-kernel durations)
#   fetch/   rocprofv3 --pmc FETCH_SIZE      (separate pass, kernel-trace only)
#   write/   rocprofv3 --pmc WRITE_SIZE      (separate pass)
#   calib/   the same two counters over a torch reduction of a known 1 GiB buffer (byte-count calibration)
# profiles/summarize.py turns them into the committed text summary.
set -u
TAG=${1:-r01}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- $BENCH > $OUT/trace.json 2> $OUT/trace.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o bench -- $BENCH > $OUT/fetch.json 2> $OUT/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o bench -- $BENCH > $OUT/write.json 2> $OUT/write.err
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
find $OUT -name "*.csv" | head -40
# keep the merge small: drop the raw sqlite/large files, keep csv
find $OUT -name "*.db" -delete
du -sh $OUT
