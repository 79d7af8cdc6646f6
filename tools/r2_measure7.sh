#!/bin/bash
cd /root/repo
O=gpurun_out; L=$O/ab7.log; : > $L
run() { hs=$1; lw=$2; geom=$3; MC_LIN_WAVES=$lw DBGS=0 timeout -k 10 120 python3 tools/gemv_ab.py $hs $geom >> $L 2>> $O/ab7.err || echo "{\"hsaco\": \"$hs\", \"failed\": $?}" >> $L; }
run tools/variants/dec.hsaco 8 512x1
run tools/variants/dec_r4q.hsaco 8 512x1
cat $L
