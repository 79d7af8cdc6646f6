#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): produces the rocprofv3 evidence for one round.
#   profiles-style outputs are written to gpurun_out/ and copied into profiles/ by hand.
# usage: tools/profile_round.sh r01 [bench args...]
set -u
R=${1:-r01}; shift || true
OUT=/root/repo/gpurun_out
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
# hipGraph replay under rocprofv3 segfaults with the ROCm 7.2 runtime on this image (reproduced with
# --kernel-trace alone), so the profiled runs launch the same kernels eagerly (--no-graph)
ARGS="--steps 64 --warmup 8 --no-cpu-baseline --no-graph --no-other-configs $*"
# 1. kernel durations
rm -rf /tmp/p_trace; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_trace -- python3 /root/repo/bench.py $ARGS > $OUT/${R}_bench_under_trace.json 2> /tmp/p_trace.err
cp $(find /tmp/p_trace -name "*kernel_stats.csv" | head -1) $OUT/${R}_kernel_stats.csv
# 2/3. HBM traffic counters, one pass each (TCC slots: FETCH_SIZE 3, WRITE_SIZE 2)
# (bench.py itself crashes rocprofv3 --pmc on this image -- device-to-host copies under counter
#  collection -- so the counter passes attach to tools/pmc_gemv.py: the same decoder, the same
#  kernels and weights, launched through mc_decoder_time_gemv)
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/p_$C; rocprofv3 --pmc $C --output-format csv -d /tmp/p_$C -- python3 /root/repo/tools/pmc_gemv.py > /dev/null 2> /tmp/p_$C.err
done
cd /root/repo
python3 tools/pmc_traffic.py /tmp/p_FETCH_SIZE /tmp/p_WRITE_SIZE > $OUT/${R}_pmc_traffic.json
cat $OUT/${R}_pmc_traffic.json
head -12 $OUT/${R}_kernel_stats.csv | cut -c1-140
