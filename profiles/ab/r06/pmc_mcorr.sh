#!/bin/bash
# SQ counters of the mcorr kernel under one command:  bash profiles/ab/r06/pmc_mcorr.sh <tag> <env assignments / command...>
TAG=$1; shift
ROOT=/root/repo; OUT=$ROOT/gpurun_out/r06/pmc_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU --output-format csv -d $OUT/p1 -o t -- env "$@" > /dev/null 2> $OUT/p1.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p2 -o t -- env "$@" > /dev/null 2> $OUT/p2.err
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/p1/**/*counter_collection.csv", recursive=True)):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if "mcorr_kernel" not in r["Kernel_Name"]: continue
        per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    byc = collections.defaultdict(list)
    for (d, c), v in per.items(): byc[c].append(v)
    out = {c: sum(v) / len(v) for c, v in byc.items()}
    w = out.get("SQ_WAVES", 1.0)
    print("$TAG", " ".join("%s=%.4g" % (c, v) for c, v in sorted(out.items())), " VALU/wave=%.1f SALU/wave=%.1f" % (out.get("SQ_INSTS_VALU", 0) / w, out.get("SQ_INSTS_SALU", 0) / w))
for f in sorted(glob.glob("$OUT/p2/**/*kernel_stats.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "mcorr_kernel" in r["Name"]: print("$TAG", "kernel avg ns", r["AverageNs"], "calls", r["Calls"])
PY
