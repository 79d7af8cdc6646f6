# Runs ON THE GPU BOX: the sampler's parity tests, then what it costs per token and its kernels' durations
cd /tmp; export TMPDIR=/tmp
python3 -m pytest /root/repo/tests/test_sampler_gpu.py -x -q > /root/repo/gpurun_out/sampler_tests.log 2>&1 || { tail -40 /root/repo/gpurun_out/sampler_tests.log; exit 1; }
tail -2 /root/repo/gpurun_out/sampler_tests.log
python3 /root/repo/tools/sampler_bench.py && python3 /root/repo/tools/sampler_bench.py 256000 64 && python3 /root/repo/tools/sampler_bench.py 128256 128
rm -rf /tmp/p_s; MC_NO_GRAPH=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_s -- python3 /root/repo/tools/sampler_bench.py > /dev/null 2> /tmp/p_s.err
grep "mc_topk\|mc_sample\|mc_argmax" $(find /tmp/p_s -name "*kernel_stats.csv" | head -1) | cut -d, -f1-4
