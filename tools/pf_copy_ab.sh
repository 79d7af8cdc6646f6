#!/bin/bash
# Tuning aid: the prompt pass of the BASELINE model shape, alternating on one box, under the environments given (default: with / without the
# dequantised bfloat16 copy of the quantised matrices, decoder.cc plain_copy_ok).   usage: tools/pf_copy_ab.sh [rounds=2] ["ENV=.. ENV=.." ...]
cd "${GRAFT_REPO_ROOT:-.}"
R=${1:-2}; shift
[ $# -eq 0 ] && set -- "MC_PF_PLAIN_COPY=1" "MC_PF_PLAIN_COPY=0"
for r in $(seq 1 "$R"); do
  for c in "$@"; do
    echo "== $c round $r"
    env $c timeout -k 10 240 python tools/prefill_bench.py 512 1024 2048 2>&1 | tail -3 || exit 1
  done
done
