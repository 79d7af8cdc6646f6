#!/bin/bash
cd /root/repo
O=gpurun_out
timeout -k 10 300 python3 -m pytest tests/test_context_gpu.py -x -q -k "folded or benchmark_context or 70b" > $O/fold_tests.log 2>&1; echo tests rc=$?; tail -5 $O/fold_tests.log
for f in 1 0; do MC_PV_FOLD=$f timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-other-configs --steps 128 --warmup 16 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench fold=$f', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], {k:round(v['avg_launch_us'],2) for k,v in d['roofline']['other_gemvs'].items()})"; done
