#!/bin/bash
# Round-5 measurements, part 10: how long a block that takes one period per call should sleep between looks (no polling, timer slack 1 us)
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05n; mkdir -p $OUT
cd /tmp
run() { echo "== PPC=${PPC:-1} $*"; env GSH_TEST_NO_REFERENCE=1 "$@" $ROOT/tests/host/test_tracking_adapters bench 32 25000000 4000000 ${PPC:-1} 2.5 2>/dev/null | grep DROPIN_JSON | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l[len('DROPIN_JSON'):])
    print({k: d[k] for k in ('channel_periods_per_s', 'mean_record_wait_us', 'record_waits', 'block_calls', 'empty_calls', 'mean_general_work_us', 'waiting_for_the_slowest_reader_seconds') if k in d})
"; }
{
for sl in 10 12 15 20 25 30 40; do run GSH_TRK_LIVE_SLEEP_US=$sl; run GSH_TRK_LIVE_SLEEP_US=$sl; done
for ppc in 2 4 8; do for sl in 12 20; do PPC=$ppc run GSH_TRK_LIVE_SINGLE_MAX=8 GSH_TRK_LIVE_SLEEP_US=$sl; done; PPC=$ppc run GSH_X=0; done
PPC=20 run GSH_TRK_LIVE_SINGLE_MAX=64 GSH_TRK_LIVE_SLEEP_US=20
PPC=20 run GSH_X=0
} > $OUT/dropin_sleep_sweep.txt 2>&1
cat $OUT/dropin_sleep_sweep.txt
