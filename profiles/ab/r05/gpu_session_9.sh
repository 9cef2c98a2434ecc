#!/bin/bash
# Round-5 measurements, part 9: drop-in with the adaptive waiting discipline (one period per call: no polling, 5 us sleeps, timer slack 1 us), live tests, the config-4 closed loop
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05m; mkdir -p $OUT
cd /tmp
run() { echo "== $*"; env GSH_TEST_NO_REFERENCE=1 "$@" $ROOT/tests/host/test_tracking_adapters bench 32 25000000 4000000 ${PPC:-1} 2.5 2>/dev/null | grep DROPIN_JSON | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l[len('DROPIN_JSON'):])
    print({k: d[k] for k in ('channel_periods_per_s', 'mean_record_wait_us', 'record_waits', 'block_calls', 'empty_calls', 'mean_general_work_us', 'waiting_for_the_slowest_reader_seconds', 'push_seconds', 'residencies') if k in d})
"; }
{
run GSH_X=0
run GSH_X=0
run GSH_X=0
run GSH_TRK_LIVE_SLEEP_US=3
run GSH_TRK_LIVE_SLEEP_US=8
run GSH_TRK_LIVE_SLEEP_US=12
run GSH_TRK_LIVE_SPIN_US_SINGLE=5
run GSH_TEST_ROOM_WAIT=spin
PPC=20 run GSH_X=0
PPC=20 run GSH_X=0
PPC=4 run GSH_X=0
PPC=2 run GSH_X=0
} > $OUT/dropin_adaptive.txt 2>&1
cat $OUT/dropin_adaptive.txt
cd $ROOT
python -m pytest tests/test_tracking_live_gpu.py -m gpu -q -x > $OUT/live_tests.log 2>&1; tail -3 $OUT/live_tests.log
python - <<'P' > $OUT/config4.txt 2>&1
import sys, os, json
sys.path.insert(0, os.getcwd())
import torch, bench
print(json.dumps(bench.closed_loop_config4_metric(torch, 0), indent=1))
P
cat $OUT/config4.txt | head -30
