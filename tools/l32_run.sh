# Runs ON THE GPU BOX: the reference's default model (Llama-3.2-1B, bf16 weights) -- tokens/s and the kernels of its layer
cd /tmp; export TMPDIR=/tmp
python3 /root/repo/bench.py --model llama3.2-1b --wbits 16 --steps 256 --warmup 32 --no-cpu-baseline --no-other-configs --no-roofline 2>/dev/null | tail -n 1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('llama3.2-1b bf16:', round(d['value'],1), 'tokens/s', round(d['ms_per_step']*1e3/16,2), 'us per layer incl. head share')"
rm -rf /tmp/p_l; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_l -- python3 /root/repo/bench.py --model llama3.2-1b --wbits 16 --steps 64 --warmup 8 --no-cpu-baseline --no-graph --no-other-configs --no-roofline > /dev/null 2> /tmp/p_l.err
f=$(find /tmp/p_l -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f /root/repo/gpurun_out/r04_kernel_stats_llama32_1b.csv && head -9 $f | cut -d, -f1-4
