# Runs ON THE GPU BOX: kernels of short prompts (rows given as arguments, default 32 64)
cd /tmp; export TMPDIR=/tmp
for n in ${@:-32 64}; do
  rm -rf /tmp/pf; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf -- python3 /root/repo/tools/prefill_bench.py $n > /dev/null 2>&1
  f=$(find /tmp/pf -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f /root/repo/gpurun_out/r04_prefill_kernel_stats_$n.csv && echo "== $n rows" && grep -v "synth\|fillBuffer\|repack" $f | head -9 | cut -d, -f1-4
done
