#!/bin/bash
# GEMV structure A/B, batch 1 (runs on the GPU box)
cd /root/repo
O=gpurun_out/ab1.log; : > $O
run() { v=$1; shift; timeout 200 python3 tools/gemv_ab.py tools/variants/$v.hsaco "$@" >> $O 2>> gpurun_out/ab1.err || echo "{\"hsaco\": \"$v\", \"failed\": $?}" >> $O; }
run orig 256x2
for v in wm wm_x wm_r8p8 wm_r8p8x wm_r8p8xi wm_r6p6xi wm_r4p4xi wm_r8p8xi_nt wm_r8p1x; do run $v 256x2f 512x1f; done
for v in wm_r3p3xi_lb1024 wm_r3p1x_lb1024; do run $v 1024x1f 512x1f 512x2f; done
cat $O
