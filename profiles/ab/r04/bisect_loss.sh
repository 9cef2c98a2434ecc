# the loss-of-lock case of tests/host/test_tracking_adapters with several builds of the library put in place one after the other (build/variants/lib_<tag>.so)
cd /root/repo; cp gnss-sdr_amd/libgnss_sdr_hip.so /tmp/lib_cur.so
for v in "$@" cur; do
  if [ $v = cur ]; then cp /tmp/lib_cur.so gnss-sdr_amd/libgnss_sdr_hip.so; else cp build/variants/lib_$v.so gnss-sdr_amd/libgnss_sdr_hip.so; fi
  echo "== $v: $(cd /tmp && timeout 60 /root/repo/tests/host/test_tracking_adapters loss 2>&1 | grep 'noise only\|OK$\|failure' | tr '\n' ' ')"
done
cp /tmp/lib_cur.so gnss-sdr_amd/libgnss_sdr_hip.so
