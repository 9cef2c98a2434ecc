#!/bin/bash
# A/B of the drop-in leg's push policy (round 4), one box, back to back, reference comparison off: GSH_TRK_PUSH_TRY x GSH_TRK_PUSH_BATCH x periods per call
cd /tmp
for rep in 1 2; do
for ppc in 1 20; do
for try in 0 1; do
for batch in "" 2 4; do
  line=$(env GSH_TEST_NO_REFERENCE=1 GSH_TRK_PUSH_TRY=$try ${batch:+GSH_TRK_PUSH_BATCH=$batch} timeout 120 /root/repo/tests/host/test_tracking_adapters bench 32 25000000 400000 $ppc 2.0 2>&1 | grep DROPIN_JSON | sed 's/DROPIN_JSON//')
  echo "rep $rep ppc $ppc try $try batch ${batch:-0}: $(echo "$line" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['channel_periods_per_s']), 'c-p/s; record waits', d['record_waits'], 'x', d['mean_record_wait_us'], 'us; calls', d['block_calls'], 'empty', d['empty_calls'], 'gw', d['mean_general_work_us'], 'us; slowest-wait', d['waiting_for_the_slowest_reader_seconds'], 's; push', d['push_seconds'], 's', d['push_GBs'], 'GB/s')")"
done; done; done; done
