#!/bin/bash
cd /root/repo
O=gpurun_out/ab5.log; : > $O
run() { v=$1; shift; timeout 200 python3 tools/gemv_ab.py tools/variants/$v.hsaco "$@" >> $O 2>> gpurun_out/ab5.err || echo "{\"hsaco\": \"$v\", \"failed\": $?}" >> $O; }
export DBGS=0 MC_GEMV_LIN=1
run lin 256x2 384x2 512x1
run lin_x 256x2 512x1 384x1
run lin_if4x 512x1 512x2 256x4
run lin1k_if4 1024x1
run lin1k_if4x 1024x1 512x2 512x1
run lin1k_if8x 1024x1 512x1
cat $O
