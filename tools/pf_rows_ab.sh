#!/bin/bash
# Tuning aid: short prompts of the BASELINE model shape under different gates of the ping-pong GEMM, alternating on one box.
#   usage: tools/pf_rows_ab.sh "len len .." "ENV=.. ENV=.." ["ENV=.."...]
cd "${GRAFT_REPO_ROOT:-.}"
LENS=${1:-"192 256 320 383"}; shift
[ $# -eq 0 ] && set -- "MC_PF_GEMM8_ROWS=384" "MC_PF_GEMM8_ROWS=256" "MC_PF_GEMM8_ROWS=192"
for r in 1 2; do for c in "$@"; do echo "== $c round $r"; env $c timeout -k 10 240 python tools/prefill_bench.py $LENS 2>&1 | tail -$(echo $LENS | wc -w) || exit 1; done; done
