# Runs ON THE GPU BOX: the Gemma-7B-shaped leg -- parity tests of what it launches, then tokens/s (graph replay) and kernel stats
cd /tmp; export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest /root/repo/tests/test_context_gpu.py /root/repo/tests/test_full_size_gpu.py /root/repo/tests/test_attn_kernels_gpu.py -x -q -k "gemma or one_launch" > /root/repo/gpurun_out/gm_tests.log 2>&1 || { tail -30 /root/repo/gpurun_out/gm_tests.log; exit 1; }
tail -2 /root/repo/gpurun_out/gm_tests.log
for i in 1 2 3; do CASE=gemma MC_SKIP_FILL=1 python3 /root/repo/tools/configs_run.py; done
rm -rf /tmp/p_c; CASE=gemma MC_NO_GRAPH=1 MC_SKIP_FILL=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c -- python3 /root/repo/tools/configs_run.py > /root/repo/gpurun_out/r04_config_gemma.json 2> /tmp/p_c.err
f=$(find /tmp/p_c -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f /root/repo/gpurun_out/r04_kernel_stats_gemma.csv && head -8 $f | cut -d, -f1-4
