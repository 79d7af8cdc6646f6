#!/bin/bash
cd /root/repo
O=gpurun_out/ab3.log; : > $O
run() { v=$1; shift; timeout 200 python3 tools/gemv_ab.py tools/variants/$v.hsaco "$@" >> $O 2>> gpurun_out/ab3.err || echo "{\"hsaco\": \"$v\", \"failed\": $?}" >> $O; }
export MC_GEMV_LIN=1 DBGS=0
for v in lin_noload lin_a4_noload; do run $v 256x2 256x4 256x1; done
cat $O
