#!/bin/bash
cd /root/repo
O=gpurun_out
for v in only4 only14; do MC_HSACO=tools/variants/$v.hsaco timeout -k 10 200 python3 -m pytest tests/test_context_gpu.py -x -q -k "70b" > $O/t70_$v.log 2>&1; echo "$v rc=$?"; grep -n "AssertionError" $O/t70_$v.log | head -2; done
timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-other-configs --steps 128 --warmup 16 2>>$O/ab23.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], {k:round(v['avg_launch_us'],2) for k,v in d['roofline']['other_gemvs'].items()})"
