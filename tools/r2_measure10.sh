#!/bin/bash
cd /root/repo
O=gpurun_out; L=$O/ab10.log; : > $L
run() { hs=$1; lw=$2; geom=$3; MC_LIN_WAVES=$lw DBGS=0 timeout -k 10 120 python3 tools/gemv_ab.py $hs $geom >> $L 2>> $O/ab10.err || echo "{\"hsaco\": \"$hs\", \"failed\": $?}" >> $L; }
run metalchat_amd/lib/metalchat.hsaco 8 512x1
run tools/variants/w12r2.hsaco 12 768x1
run tools/variants/w12r4.hsaco 12 768x1
run tools/variants/w12tp2.hsaco 12 768x1
run tools/variants/w8nt0.hsaco 8 512x1
run tools/variants/w6r4.hsaco 6 384x1
run tools/variants/w16tp2.hsaco 16 1024x1
cat $L
