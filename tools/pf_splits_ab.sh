#!/bin/bash
# Runs ON THE GPU BOX: the prompt pass at 512 / 2048 rows under several split-K limits of the ping-pong GEMM, alternating on one box
for i in 1 2; do
  for V in "MC_PF_GEMM8_MAXSPLIT=16" "MC_PF_GEMM8_MAXSPLIT=4" "MC_PF_GEMM8_MAXSPLIT=2" "MC_PF_GEMM8_MINKT=16" "MC_PF_GEMM8_MINKT=4"; do
    echo "== $V"; env $V python3 tools/prefill_bench.py 512 2048 2>/dev/null
  done
done
