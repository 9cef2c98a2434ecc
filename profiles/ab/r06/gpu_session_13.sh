#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r06; export TMPDIR=/tmp
{
for N in "128000 32e6" "50000 50e6"; do
cd /tmp; rm -rf /tmp/kt; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o t -- python /root/repo/profiles/ab/r06/acq_one.py $N > /tmp/kt.log 2>&1
grep "ms per batch" /tmp/kt.log
python3 - <<'PY'
import csv, glob
for f in glob.glob("/tmp/kt/**/*kernel_stats.csv", recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:6]:
        print("  %-110s calls %5s avg %9.1f us  %5.1f %%" % (r["Name"][:110], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
cd /root/repo
done
} > gpurun_out/r06/session13.txt 2>&1
grep -v amdgpu.ids gpurun_out/r06/session13.txt | tail -30
