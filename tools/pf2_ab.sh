cd /root/repo
for T in 1 2 3; do echo "== MC_PF2_WGS_PER_CU=$T"; MC_PF2_WGS_PER_CU=$T python3 tools/prefill_bench.py 8 16 32 64; done
