#!/bin/bash
# Tuning aid: the prompt pass of the BASELINE model shape with / without the dequantised bfloat16 copy of the quantised matrices
# (decoder.cc plain_copy_ok, MC_PF_PLAIN_COPY), alternating on one box.   usage: tools/pf_copy_ab.sh [rounds=2]
cd "${GRAFT_REPO_ROOT:-.}"
for r in $(seq 1 "${1:-2}"); do
  for c in 1 0; do
    echo "== MC_PF_PLAIN_COPY=$c round $r"
    MC_PF_PLAIN_COPY=$c timeout -k 10 240 python tools/prefill_bench.py 512 1024 2048 2>&1 | tail -3 || exit 1
  done
done
