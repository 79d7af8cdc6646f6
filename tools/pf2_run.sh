# Runs ON THE GPU BOX: the prompt-pass tests, then times per prompt length and the kernels of an 8-row and a 512-row prompt
cd /tmp; export TMPDIR=/tmp
python3 -m pytest /root/repo/tests/test_prefill_gpu.py -x -q > /root/repo/gpurun_out/t_pf2.log 2>&1 || { tail -40 /root/repo/gpurun_out/t_pf2.log; exit 1; }
tail -2 /root/repo/gpurun_out/t_pf2.log
python3 /root/repo/tools/prefill_bench.py 8 16 32 64 65 128 256 512 2048 | tee /root/repo/gpurun_out/r04_prefill_times.log
for n in 8 512; do
  rm -rf /tmp/pf; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf -- python3 /root/repo/tools/prefill_bench.py $n > /dev/null 2>&1
  f=$(find /tmp/pf -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f /root/repo/gpurun_out/r04_prefill_kernel_stats_$n.csv && head -11 $f | cut -d, -f1-4
done
