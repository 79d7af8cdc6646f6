#!/bin/bash
cd /root/repo
O=gpurun_out; L=$O/ab28.log; : > $L
run() { hs=$1; lw=$2; geom=$3; shift 3; env "$@" MC_LIN_WAVES=$lw DBGS=0 timeout -k 10 120 python3 tools/gemv_ab.py $hs $geom >> $L 2>> $O/ab28.err || echo "{\"hsaco\": \"$hs\", \"failed\": $?}" >> $L; }
for args in "8192 16384 0" "4096 28672 1" "14336 8192 0"; do timeout -k 10 120 python3 tools/lin_check.py $args 2>&1 | tail -2; done
run metalchat_amd/lib/metalchat.hsaco 8 512x1 A=1
run tools/variants/norawpark.hsaco 8 512x1 A=1
cat $L
timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-other-configs --steps 128 --warmup 16 2>>$O/ab28.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], {k:round(v['avg_launch_us'],2) for k,v in d['roofline']['other_gemvs'].items()})"
MC_HSACO=tools/variants/norawpark.hsaco timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-other-configs --steps 128 --warmup 16 2>>$O/ab28.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench norawpark', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], {k:round(v['avg_launch_us'],2) for k,v in d['roofline']['other_gemvs'].items()})"
timeout -k 10 400 python3 -m pytest tests/test_context_gpu.py tests/test_golden_gpu.py -x -q > $O/rawpark_tests.log 2>&1; echo tests rc=$?; tail -2 $O/rawpark_tests.log
