#!/bin/bash
# Round-5 measurements, part 16: kernel-trace timelines of pipelined batches -- do successive batches' cell kernels overlap?
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05p; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cfg in "2 28" "3 32" "4 32"; do
  set -- $cfg
  rm -rf /tmp/tl_$1_$2
  GSH_ACQ_LANES=$1 GSH_OC_WG_PER_XCD=$2 PYTHONPATH=$ROOT:$ROOT/tests rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$1_$2 -o t -- python $ROOT/profiles/ab/r05/acq_long.py 300 1 > $OUT/tl_$1_$2.run 2>&1
  { echo "== $1 lanes, $2 work-groups per XCD"; tail -1 $OUT/tl_$1_$2.run; python $ROOT/profiles/ab/r05/acq_timeline.py /tmp/tl_$1_$2; } > $OUT/timeline_$1_$2.txt 2>&1
  cat $OUT/timeline_$1_$2.txt
done
