#!/bin/bash
cd /root/repo
O=gpurun_out; L=$O/ab27.log; : > $L
run() { hs=$1; lw=$2; geom=$3; shift 3; env "$@" MC_LIN_WAVES=$lw DBGS=0 timeout -k 10 120 python3 tools/gemv_ab.py $hs $geom >> $L 2>> $O/ab27.err || echo "{\"hsaco\": \"$hs\", \"failed\": $?}" >> $L; }
run metalchat_amd/lib/metalchat.hsaco 8 512x1 A=1
run tools/variants/ldsr_on.hsaco 8 512x1 MC_LIN_LDS_RING=1
run tools/variants/ldsr_nosc.hsaco 8 512x1 MC_LIN_LDS_RING=1
run tools/variants/ldsr_nored.hsaco 8 512x1 MC_LIN_LDS_RING=1
cat $L
