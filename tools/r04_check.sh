# Runs ON THE GPU BOX: the whole GPU suite, then the kernel durations of the headline token with wq|wk|wv as a launch of its own
# (MC_ATTN_QKV=0: the stand-alone attention + Wo launch with this round's hand-offs)
cd /tmp; export TMPDIR=/tmp
python3 -m pytest /root/repo/tests -m gpu -x -q > /root/repo/gpurun_out/t_all.log 2>&1 || { tail -30 /root/repo/gpurun_out/t_all.log; exit 1; }
tail -2 /root/repo/gpurun_out/t_all.log
rm -rf /tmp/p_q; MC_ATTN_QKV=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_q -- python3 /root/repo/bench.py --steps 64 --warmup 8 --no-cpu-baseline --no-graph --no-other-configs --no-roofline > /root/repo/gpurun_out/r04_bench_qkv0_under_trace.json 2> /tmp/p_q.err
cp $(find /tmp/p_q -name "*kernel_stats.csv" | head -1) /root/repo/gpurun_out/r04_kernel_stats_attn_qkv_off.csv
head -8 /root/repo/gpurun_out/r04_kernel_stats_attn_qkv_off.csv | cut -d, -f1-4
