#!/bin/bash
cd /root/repo
O=gpurun_out; L=$O/ab4.log; : > $L
run() { hs=$1; lw=$2; geom=$3; MC_LIN_WAVES=$lw DBGS=0 timeout -k 10 120 python3 tools/gemv_ab.py $hs $geom >> $L 2>> $O/ab4.err || echo "{\"hsaco\": \"$hs\", \"failed\": $?}" >> $L; }
run metalchat_amd/lib/metalchat.hsaco 8 512x1
run tools/variants/r4q.hsaco 8 512x1
run tools/variants/r8q.hsaco 8 512x1
run tools/variants/r2q.hsaco 8 512x1
run tools/variants/w16r4q.hsaco 16 1024x1
run tools/variants/w16r2q.hsaco 16 1024x1
run tools/variants/w4r4q.hsaco 4 256x2
run tools/variants/w4r8q.hsaco 4 256x2
run tools/variants/w4r8q.hsaco 4 256x1
cat $L
