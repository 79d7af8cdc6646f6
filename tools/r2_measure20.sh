#!/bin/bash
cd /root/repo
O=gpurun_out; L=$O/ab20.log; : > $L
run() { hs=$1; lw=$2; geom=$3; MC_LIN_WAVES=$lw DBGS=0 timeout -k 10 120 python3 tools/gemv_ab.py $hs $geom >> $L 2>> $O/ab20.err || echo "{\"hsaco\": \"$hs\", \"failed\": $?}" >> $L; }
run metalchat_amd/lib/metalchat.hsaco 8 512x1
MC_TIME_GEMV_LAYERS=1 run metalchat_amd/lib/metalchat.hsaco 8 512x1
MC_TIME_GEMV_LAYERS=2 run metalchat_amd/lib/metalchat.hsaco 8 512x1
cat $L
