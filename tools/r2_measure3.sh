#!/bin/bash
cd /root/repo
O=gpurun_out; L=$O/ab3.log; : > $L
run() { hs=$1; lw=$2; geom=$3; MC_LIN_WAVES=$lw DBGS=0 timeout -k 10 120 python3 tools/gemv_ab.py $hs $geom >> $L 2>> $O/ab3.err || echo "{\"hsaco\": \"$hs\", \"failed\": $?}" >> $L; }
run metalchat_amd/lib/metalchat.hsaco 8 512x1
run tools/variants/r4.hsaco 8 512x1
run tools/variants/r8.hsaco 8 512x1
run tools/variants/r2.hsaco 8 512x1
run tools/variants/w16r4.hsaco 16 1024x1
run tools/variants/w16r2.hsaco 16 1024x1
run tools/variants/w4r4.hsaco 4 256x2
run tools/variants/w4r8.hsaco 4 256x2
run tools/variants/w4r8.hsaco 4 256x1
cat $L
MC_HSACO=tools/variants/r4.hsaco timeout -k 10 300 python3 -m pytest tests/test_gemv_gpu.py tests/test_context_gpu.py -x -q > $O/r4_tests.log 2>&1; echo tests rc=$?; tail -3 $O/r4_tests.log
