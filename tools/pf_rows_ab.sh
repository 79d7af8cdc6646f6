cd "${GRAFT_REPO_ROOT:-.}"
for r in 1 2; do for c in "MC_PF_GEMM8_ROWS=384" "MC_PF_GEMM8_ROWS=256" "MC_PF_GEMM8_ROWS=192"; do echo "== $c round $r"; env $c timeout -k 10 240 python tools/prefill_bench.py 192 256 320 383 2>&1 | tail -4 || exit 1; done; done
