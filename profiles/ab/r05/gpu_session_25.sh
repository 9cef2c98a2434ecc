#!/bin/bash
# Round-5 measurements, part 25: the closed-loop period's phases with the seed tables in (-DGSH_TRK_PROFILE=2: correlation phase; =1: serial section)
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05p; mkdir -p $OUT
{
GSH_LIB_PATH=$ROOT/build/variants/lib_trk_prof2.so GSH_PHASE_DETAIL=2 GSH_LOOP_AB_CONF=lock python profiles/ab/closed_loop_phases.py 2>/dev/null | tail -2
GSH_LIB_PATH=$ROOT/build/variants/lib_trk_prof1.so GSH_PHASE_DETAIL=1 GSH_LOOP_AB_CONF=lock python profiles/ab/closed_loop_phases.py 2>/dev/null | tail -2
} > $OUT/closed_loop_phases_seed_tables.txt 2>&1
cat $OUT/closed_loop_phases_seed_tables.txt
