#!/bin/bash
# Runs ON THE GPU BOX: kernel durations of one prompt length grouped by (kernel, grid, workgroup) -- tells the split-K launches of one kernel name apart.
#   usage: tools/prefill_trace_groups.sh 512 [ENV=..]
N=${1:-512}; shift
cd /tmp; export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
rm -rf /tmp/p_pfg
rocprofv3 --kernel-trace --output-format csv -d /tmp/p_pfg -- python3 /root/repo/tools/prefill_bench.py $N > /dev/null 2> /tmp/p_pfg.err
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/p_pfg/**/*kernel_trace.csv', recursive=True)[0]
g = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    key = (r['Kernel_Name'], r.get('Grid_Size_X', r.get('Grid_Size', '')), r.get('Grid_Size_Y', ''), r.get('Grid_Size_Z', ''), r.get('Workgroup_Size_X', r.get('Workgroup_Size', '')))
    g[key].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for k, v in sorted(g.items(), key=lambda kv: -sum(kv[1])):
    if k[0].startswith('mc_pf') or k[0].startswith('mc_gemv'):
        print(f"{k[0]:40s} grid {k[1]:>7s} x {k[2]:>4s} x {k[3]:>3s} wg {k[4]:>4s}  calls {len(v):4d}  avg {sum(v)/len(v)/1e3:8.2f} us")
PY
