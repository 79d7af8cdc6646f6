#!/bin/bash
cd /tmp; export TMPDIR=/tmp
O=/root/repo/gpurun_out
prof() { tag=$1; shift; rm -rf /tmp/p_$tag; env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$tag -- python3 /root/repo/bench.py --steps 64 --warmup 8 --no-cpu-baseline --no-graph --no-other-configs > $O/prof_$tag.json 2> $O/prof_$tag.err; cp $(find /tmp/p_$tag -name "*kernel_stats.csv" | head -1) $O/prof_${tag}_kernel_stats.csv; head -14 $O/prof_${tag}_kernel_stats.csv | cut -d, -f1-4 ; }
prof base MC_DUMMY=1
prof pv4 MC_PV_RANGES=4 MC_PV_BLOCK=256
prof pv2 MC_PV_RANGES=2 MC_PV_BLOCK=512
cd /root/repo
python3 bench.py --no-cpu-baseline --steps 128 --warmup 16 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['other_gemvs'])"
