#!/bin/bash
cd /root/repo
O=gpurun_out/ab4.log; : > $O
run() { v=$1; shift; timeout 200 python3 tools/gemv_ab.py tools/variants/$v.hsaco "$@" >> $O 2>> gpurun_out/ab4.err || echo "{\"hsaco\": \"$v\", \"failed\": $?}" >> $O; }
export DBGS=0
MC_GEMV_LIN=1 timeout 900 python3 -m pytest tests/test_full_size_gpu.py tests/test_decode_gpu.py -m gpu -x -q 2>&1 | tail -8 > gpurun_out/ab4_tests.log
cat gpurun_out/ab4_tests.log
MC_GEMV_LIN=0 run lin 256x2
MC_GEMV_LIN=1 run lin 256x2 256x1 256x4
cat $O
