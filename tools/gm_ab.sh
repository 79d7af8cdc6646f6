cd /tmp; export TMPDIR=/tmp
for E in MC_ATTN_QKN=1 MC_ATTN_QKN=0; do
  rm -rf /tmp/p_c; env $E CASE=gemma MC_NO_GRAPH=1 MC_SKIP_FILL=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c -- python3 /root/repo/tools/configs_run.py > /dev/null 2> /tmp/p_c.err
  echo "== $E"; head -8 $(find /tmp/p_c -name "*kernel_stats.csv" | head -1) | cut -d, -f1-4
done
