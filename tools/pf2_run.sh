cd /tmp; export TMPDIR=/tmp
python3 -m pytest /root/repo/tests/test_prefill_gpu.py -x -q > /root/repo/gpurun_out/t_pf2.log 2>&1 || { tail -40 /root/repo/gpurun_out/t_pf2.log; exit 1; }
tail -2 /root/repo/gpurun_out/t_pf2.log
cd /root/repo
python3 tools/prefill_bench.py 8 64 65 128 256 512 2048
MC_PF3=0 python3 tools/prefill_bench.py 128 512 2048
cd /tmp; rm -rf /tmp/pf; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf -- python3 /root/repo/tools/prefill_bench.py 512 > /dev/null 2>&1; head -12 $(find /tmp/pf -name "*kernel_stats.csv" | head -1) | cut -d, -f1-4
