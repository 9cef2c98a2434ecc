#!/bin/bash
# SQ counters of one kernel (run on the GPU box):  bash profiles/pmc_kernel.sh <kernel-name-substring> <out-tag> [bench flags...]
# Three separate --pmc passes (kernel-trace only, as the MI355X guide prescribes); prints per-launch averages.
KERN=$1; TAG=$2; shift 2
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline $*"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/a -o t -- $B > /dev/null 2> $OUT/a.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/b -o t -- $B > /dev/null 2> $OUT/b.err
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $OUT/c -o t -- $B > /dev/null 2> $OUT/c.err
python - <<PY
import csv, glob, collections
for leg in "abc":
    for f in glob.glob("$OUT/%s/*counter_collection.csv" % leg):
        per = collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            if "$KERN" not in r["Kernel_Name"]: continue
            per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
        byc = collections.defaultdict(list)
        for (d, c), v in per.items(): byc[c].append(v)
        for c, v in sorted(byc.items()): print("$KERN", leg, c, "avg per launch = %.5g" % (sum(v) / len(v)), "n=", len(v))
PY
