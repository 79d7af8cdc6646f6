cd /tmp; export TMPDIR=/tmp
for i in 1 2 3; do
echo "qkn + Wo launch: $(CASE=gemma MC_SKIP_FILL=1 python3 /root/repo/tools/configs_run.py)"
echo "rope_kv + attention with Wo (2 per CU): $(MC_ATTN_WO_2PERCU=1 CASE=gemma MC_SKIP_FILL=1 python3 /root/repo/tools/configs_run.py)"
done
rm -rf /tmp/p_c; MC_ATTN_WO_2PERCU=1 CASE=gemma MC_NO_GRAPH=1 MC_SKIP_FILL=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c -- python3 /root/repo/tools/configs_run.py > /dev/null 2> /tmp/p_c.err
f=$(find /tmp/p_c -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 $f | cut -d, -f1-4
