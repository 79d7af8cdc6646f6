#!/bin/bash
cd /root/repo
O=gpurun_out; L=$O/ab18.log; : > $L
run() { hs=$1; lw=$2; geom=$3; MC_LIN_WAVES=$lw DBGS=0 timeout -k 10 120 python3 tools/gemv_ab.py $hs $geom >> $L 2>> $O/ab18.err || echo "{\"hsaco\": \"$hs\", \"failed\": $?}" >> $L; }
run tools/variants/noskew.hsaco 8 512x1
run tools/variants/skew.hsaco 8 512x1
run tools/variants/skew_noload.hsaco 8 512x1
run tools/variants/skew_noload_w4.hsaco 4 256x1
cat $L
