#!/bin/bash
cd /root/repo
O=gpurun_out
for args in "4096 8192 0 i8" "4096 28672 1 i8" "14336 4096 0 i8" "2048 8192 0 w" "2048 16384 1 w" "4096 8192 1 w" "5632 2048 0 w" "8192 2048 0 w" "14336 4096 0 w"; do timeout -k 10 120 python3 tools/lin_check.py $args 2>&1 | tail -3; done
timeout -k 10 500 python3 -m pytest tests/test_context_gpu.py -x -q > $O/ling_ctx.log 2>&1; echo ctx rc=$?; tail -2 $O/ling_ctx.log
for e in 1 0; do MC_GEMV_LING=$e timeout -k 10 300 python3 bench.py --model llama3.2-1b --wbits 16 --no-cpu-baseline --no-other-configs --steps 256 --warmup 32 2>>$O/ling.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('llama3.2-1b bf16 LING=$e', round(d['value'],1), d['ms_per_step'], d['roofline']['avg_launch_us'], {k:round(v['avg_launch_us'],2) for k,v in d['roofline']['other_gemvs'].items()})"; done
for e in 1 0; do MC_GEMV_LING=$e timeout -k 10 300 python3 bench.py --wbits 8 --seq-len 8192 --no-cpu-baseline --no-other-configs --steps 64 --warmup 16 2>>$O/ling.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('8b int8 S=8192 LING=$e', round(d['value'],1), d['ms_per_step'], d['roofline']['avg_launch_us'], {k:round(v['avg_launch_us'],2) for k,v in d['roofline']['other_gemvs'].items()})"; done
