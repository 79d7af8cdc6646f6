#!/bin/bash
cd /root/repo
O=gpurun_out; L=$O/ab8.log; : > $L
run() { hs=$1; lw=$2; geom=$3; MC_LIN_WAVES=$lw DBGS=0 timeout -k 10 120 python3 tools/gemv_ab.py $hs $geom >> $L 2>> $O/ab8.err || echo "{\"hsaco\": \"$hs\", \"failed\": $?}" >> $L; }
run metalchat_amd/lib/metalchat.hsaco 8 512x1
HIP_FORCE_DEV_KERNARG=1 run metalchat_amd/lib/metalchat.hsaco 8 512x1
HIP_FORCE_DEV_KERNARG=0 run metalchat_amd/lib/metalchat.hsaco 8 512x1
run tools/variants/kpre.hsaco 8 512x1
HIP_FORCE_DEV_KERNARG=1 run tools/variants/kpre.hsaco 8 512x1
cat $L
for e in 0 1; do HIP_FORCE_DEV_KERNARG=$e timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 128 --warmup 16 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('KERNARG=$e', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])"; done
MC_HSACO=tools/variants/kpre.hsaco HIP_FORCE_DEV_KERNARG=1 timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 128 --warmup 16 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('kpre+KERNARG=1', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])"
