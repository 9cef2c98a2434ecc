#!/bin/bash
# Run ON THE GPU BOX from the repo root (gpurun): round-5 measurements, part 1 -- acquisition cell phases + plan-shape A/B, drop-in at one period per call under
# different waiting disciplines.  Everything lands in gpurun_out/r05e/.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05e; mkdir -p $OUT
cd $ROOT
GSH_LIB_PATH=$ROOT/build/variants/lib_ocprof.so python profiles/ab/r05/oc_cell_phases.py > $OUT/oc_cell_phases.txt 2> $OUT/oc_cell_phases.err
for v in current p402525 p254025 ex32; do
  if [ $v = current ]; then python profiles/ab/acq_ab.py; else GSH_LIB_PATH=$ROOT/build/variants/lib_$v.so python profiles/ab/acq_ab.py; fi
done > $OUT/acq_shapes_ab.txt 2> $OUT/acq_shapes_ab.err
# drop-in, 32 blocks, 25 Msps, ONE period per call, 2.5 s windows; no reference comparison (GSH_TEST_NO_REFERENCE) -- the rates are what is wanted here
cd /tmp
run() { echo "== $*"; env GSH_TEST_NO_REFERENCE=1 "$@" $ROOT/tests/host/test_tracking_adapters bench 32 25000000 4000000 1 2.5 2>/dev/null | grep DROPIN_JSON | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l[len('DROPIN_JSON'):])
    print({k: d[k] for k in ('channel_periods_per_s', 'mean_record_wait_us', 'record_waits', 'block_calls', 'empty_calls', 'mean_general_work_us', 'waiting_for_the_slowest_reader_seconds', 'push_seconds', 'residencies') if k in d})
"; }
{
run GSH_TEST_ROOM_WAIT=spin
run GSH_TEST_ROOM_WAIT=spin
run GSH_TEST_ROOM_WAIT=sleep
run GSH_TEST_ROOM_WAIT=sleep
run GSH_TEST_ROOM_WAIT=sleep GSH_TRK_TIMER_SLACK_NS=1000
run GSH_TEST_ROOM_WAIT=sleep GSH_TRK_TIMER_SLACK_NS=1000 GSH_TRK_LIVE_SLEEP_US=5
run GSH_TEST_ROOM_WAIT=sleep GSH_TRK_TIMER_SLACK_NS=1000 GSH_TRK_LIVE_SPIN_US=10 GSH_TRK_LIVE_SLEEP_US=5
run GSH_TEST_ROOM_WAIT=sleep GSH_TRK_LIVE_SPIN_US=100
run GSH_TEST_ROOM_WAIT=sleep GSH_TRK_LIVE_SPIN_US=0 GSH_TRK_TIMER_SLACK_NS=1000 GSH_TRK_LIVE_SLEEP_US=5
run GSH_TEST_ROOM_WAIT=sleep GSH_TRK_PUSH_BATCH=1
run GSH_TEST_ROOM_WAIT=sleep GSH_TRK_PUSH_BATCH=3
} > $OUT/dropin_wait_ab.txt 2>&1
# 20 periods per call, for the record
{
run20() { echo "== $* (20 periods per call)"; env GSH_TEST_NO_REFERENCE=1 "$@" $ROOT/tests/host/test_tracking_adapters bench 32 25000000 4000000 20 2.5 2>/dev/null | grep -o '"channel_periods_per_s": [0-9.]*'; }
run20 GSH_TEST_ROOM_WAIT=spin
run20 GSH_TEST_ROOM_WAIT=sleep
run20 GSH_TEST_ROOM_WAIT=sleep GSH_TRK_TIMER_SLACK_NS=1000 GSH_TRK_LIVE_SLEEP_US=5
} >> $OUT/dropin_wait_ab.txt 2>&1
cd $ROOT
python -m pytest tests/test_tracking_live_gpu.py tests/test_symbol_sync.py -m gpu -q -x > $OUT/live_tests.log 2>&1
tail -3 $OUT/live_tests.log
cat $OUT/oc_cell_phases.txt $OUT/acq_shapes_ab.txt $OUT/dropin_wait_ab.txt
