#!/bin/bash
# launches in flight: one stream against two / three streams with a bank each
cd /root/repo; mkdir -p gpurun_out/r06
timeout 600 python profiles/ab/r06/mcorr_two_streams.py 2>&1 | grep -v amdgpu > gpurun_out/r06/session30.txt
cat gpurun_out/r06/session30.txt
