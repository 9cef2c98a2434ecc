#!/bin/bash
# (a) 50 000 points as a decimation-in-time split (S = 2) now that Z is non-temporal and the combine launch has no hand-off; (b) kernel trace of one 128 000-point batch
cd /root/repo
mkdir -p gpurun_out/r06
{
echo "== 50 000 points, shipped (decimation in frequency, S = 2)"; python profiles/ab/r06/acq_128k.py 50000:50e6 2>&1 | grep "^N ="
echo "== 50 000 points, decimation in time from S = 2"; GSH_OC_DIT_MIN_S=2 python profiles/ab/r06/acq_128k.py 50000:50e6 2>&1 | grep "^N ="
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr128 -o t -- python /root/repo/profiles/ab/r06/acq_one.py 128000 32e6 > /dev/null 2>&1
f=$(find /tmp/tr128 -name "*kernel_stats.csv" | head -1); echo "== kernel trace, 128 000 points ($f)"; head -8 "$f" | cut -c1-260
} > /root/repo/gpurun_out/r06/session48.txt 2>&1
cat /root/repo/gpurun_out/r06/session48.txt
