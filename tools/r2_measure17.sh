#!/bin/bash
cd /root/repo
O=gpurun_out; L=$O/ab17.log; : > $L
run() { hs=$1; lw=$2; geom=$3; MC_LIN_WAVES=$lw DBGS=0 timeout -k 10 120 python3 tools/gemv_ab.py $hs $geom >> $L 2>> $O/ab17.err || echo "{\"hsaco\": \"$hs\", \"failed\": $?}" >> $L; }
run metalchat_amd/lib/metalchat.hsaco 8 512x1
run tools/variants/w6.hsaco 6 384x1
run tools/variants/w12.hsaco 12 768x1
cat $L
