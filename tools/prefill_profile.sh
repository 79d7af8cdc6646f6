#!/bin/bash
# Runs ON THE GPU BOX: kernel durations of one prompt length through rocprofv3 (eager launches).  usage: tools/prefill_profile.sh r06 512 [env...]
R=${1:-r06}; N=${2:-512}; shift $(( $# < 2 ? $# : 2 ))
OUT=/root/repo/gpurun_out; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
rm -rf /tmp/p_pf
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_pf -- python3 /root/repo/tools/prefill_bench.py $N > $OUT/${R}_prefill_${N}_under_trace.log 2> /tmp/p_pf.err
cp $(find /tmp/p_pf -name "*kernel_stats.csv" | head -1) $OUT/${R}_prefill_kernel_stats_${N}.csv
head -25 $OUT/${R}_prefill_kernel_stats_${N}.csv | cut -d, -f1-5
