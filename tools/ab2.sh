#!/bin/bash
# linear-order GEMV: parity first, then timing (runs on the GPU box)
cd /root/repo
O=gpurun_out/ab2.log; : > $O
MC_GEMV_LIN=1 timeout 900 python3 -m pytest tests/test_full_size_gpu.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/ab2_tests.log
run() { v=$1; shift; timeout 200 python3 tools/gemv_ab.py tools/variants/$v.hsaco "$@" >> $O 2>> gpurun_out/ab2.err || echo "{\"hsaco\": \"$v\", \"failed\": $?}" >> $O; }
DBGS=0,1 MC_GEMV_LIN=0 run lin 256x2
export MC_GEMV_LIN=1 DBGS=0
for v in lin lin_stream lin_nt0 lin_if4 lin_if12 lin_if16; do run $v 256x2 256x1 512x1 256x4; done
cat gpurun_out/ab2_tests.log; cat $O
