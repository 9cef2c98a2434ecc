#!/bin/bash
# the drop-in leg at 1 and 20 periods per call, reference comparison off, two repetitions (round 4)
cd /tmp
for rep in 1 2; do
for ppc in 1 20; do
  line=$(env GSH_TEST_NO_REFERENCE=1 "$@" timeout 120 /root/repo/tests/host/test_tracking_adapters bench 32 25000000 400000 $ppc 2.0 2>&1 | grep DROPIN_JSON | sed 's/DROPIN_JSON//')
  echo "rep $rep ppc $ppc $*: $(echo "$line" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['channel_periods_per_s']), 'c-p/s; record waits', d['record_waits'], 'x', d['mean_record_wait_us'], 'us; calls', d['block_calls'], 'empty', d['empty_calls'], 'gw', d['mean_general_work_us'], 'us; slowest-wait', d['waiting_for_the_slowest_reader_seconds'], 's; push', d['push_seconds'], 's', d['push_GBs'], 'GB/s')")"
done; done
