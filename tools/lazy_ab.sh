cd /tmp; export TMPDIR=/tmp
for E in MC_LAZY_PICK=1 MC_LAZY_PICK=0; do
rm -rf /tmp/p_l; env $E rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_l -- python3 /root/repo/bench.py --steps 64 --warmup 8 --no-cpu-baseline --no-graph --no-other-configs --no-roofline > /dev/null 2> /tmp/p_l.err
echo "== $E"; grep "mc_embed\|mc_argmax_keys\|lin2_p1_e5\|attn_qkv" $(find /tmp/p_l -name "*kernel_stats.csv" | head -1) | cut -d, -f1-4
done
cd /root/repo; bash tools/ab_env.sh 3 "MC_LAZY_PICK=0" "MC_LAZY_PICK=1" --steps 256 --warmup 32 --no-roofline
