#!/bin/bash
cd /root/repo
O=gpurun_out; L=$O/ab6.log; : > $L
run() { hs=$1; lw=$2; geom=$3; MC_LIN_WAVES=$lw DBGS=0 timeout -k 10 120 python3 tools/gemv_ab.py $hs $geom >> $L 2>> $O/ab6.err || echo "{\"hsaco\": \"$hs\", \"failed\": $?}" >> $L; }
run tools/variants/w4_noload.hsaco 4 256x1
run tools/variants/w4_noload.hsaco 4 256x2
run tools/variants/w8_noload.hsaco 8 512x1
run tools/variants/w8_noload_a2.hsaco 8 512x1
run tools/variants/w4_noload_a2.hsaco 4 256x1
run tools/variants/w8_a2.hsaco 8 512x1
run tools/variants/w8_a4.hsaco 8 512x1
cat $L
