#!/bin/bash
cd /root/repo
O=gpurun_out
timeout -k 10 400 python3 -m pytest tests/test_gemv_gpu.py -x -q > $O/ldsr_tests1.log 2>&1; echo gemv tests rc=$?; tail -3 $O/ldsr_tests1.log
L=$O/ab21.log; : > $L
run() { hs=$1; lw=$2; geom=$3; MC_LIN_WAVES=$lw DBGS=0 timeout -k 10 120 python3 tools/gemv_ab.py $hs $geom >> $L 2>> $O/ab21.err || echo "{\"hsaco\": \"$hs\", \"failed\": $?}" >> $L; }
run metalchat_amd/lib/metalchat.hsaco 8 512x1
cat $L
timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-other-configs --steps 128 --warmup 16 2>>$O/ab21.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], {k:round(v['avg_launch_us'],2) for k,v in d['roofline']['other_gemvs'].items()})"
timeout -k 10 600 python3 -m pytest tests/test_decode_gpu.py tests/test_context_gpu.py tests/test_golden_gpu.py tests/test_full_size_gpu.py -x -q > $O/ldsr_tests2.log 2>&1; echo tests rc=$?; tail -3 $O/ldsr_tests2.log
