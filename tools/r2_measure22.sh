#!/bin/bash
cd /root/repo
O=gpurun_out
for e in "MC_PV_FOLD=1" "MC_PV_FOLD=0"; do env $e timeout -k 10 200 python3 -m pytest tests/test_context_gpu.py -x -q -k "70b" > $O/t70_$e.log 2>&1; echo "$e rc=$?"; grep -n "AssertionError" $O/t70_$e.log | head -2; done
L=$O/ab22.log; : > $L
run() { hs=$1; lw=$2; geom=$3; MC_LIN_WAVES=$lw DBGS=0 timeout -k 10 120 python3 tools/gemv_ab.py $hs $geom >> $L 2>> $O/ab22.err || echo "{\"hsaco\": \"$hs\", \"failed\": $?}" >> $L; }
run metalchat_amd/lib/metalchat.hsaco 8 512x1
cat $L
timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-other-configs --steps 128 --warmup 16 2>>$O/ab22.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], {k:round(v['avg_launch_us'],2) for k,v in d['roofline']['other_gemvs'].items()})"
