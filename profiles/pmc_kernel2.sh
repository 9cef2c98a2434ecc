#!/bin/bash
# SQ / TA / TCP counters of one kernel (run on the GPU box):  bash profiles/pmc_kernel2.sh <kernel-substring> <out-tag> <cmd...>
# Separate --pmc passes with --kernel-trace only (MI355X guide); prints per-launch averages and writes them to gpurun_out/pmc_<tag>/summary.txt
KERN=$1; TAG=$2; shift 2
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="$*"
i=0
while read -r SET; do
  [ -z "$SET" ] && continue
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/p$i -o t -- $CMD > /dev/null 2> $OUT/p$i.err
done <<SETS
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
GRBM_GUI_ACTIVE GRBM_COUNT SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS
SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT32 SQ_IFETCH
TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
SQ_WAVES_EQ_64 SQ_BUSY_CU_CYCLES SQ_LEVEL_WAVES SQ_ACCUM_PREV_HIRES SQC_ICACHE_MISSES SQC_ICACHE_REQ SQ_INSTS_BRANCH SQ_INSTS_WAVE32_LDS
SETS
python3 - <<PY | tee $OUT/summary.txt
import csv, glob, collections
for f in sorted(glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True)):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if "$KERN" not in r["Kernel_Name"]: continue
        per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    byc = collections.defaultdict(list)
    for (d, c), v in per.items(): byc[c].append(v)
    for c, v in sorted(byc.items()): print("$KERN", c, "avg per launch = %.5g" % (sum(v) / len(v)), "n=", len(v))
PY
