#!/bin/bash
cd /tmp; export TMPDIR=/tmp
O=/root/repo/gpurun_out
rm -rf /tmp/p_tr; rocprofv3 --kernel-trace --output-format csv -d /tmp/p_tr -- python3 /root/repo/bench.py --steps 16 --warmup 8 --no-cpu-baseline --no-graph --no-other-configs --no-roofline > $O/tr15.json 2> $O/tr15.err
f=$(find /tmp/p_tr -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# keep the last 8 tokens' worth: find decode kernels only
names=collections.defaultdict(list)
seq=[]
for r in rows:
    n=r['Kernel_Name']; d=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
    seq.append((n,d,int(r['Start_Timestamp']),int(r['End_Timestamp'])))
# last 2000 kernels
tail=seq[-1600:]
import statistics
by=collections.defaultdict(list)
gaps=collections.defaultdict(list)
for i,(n,d,s,e) in enumerate(tail):
    by[n].append(d)
    if i>0: gaps[n].append(s-tail[i-1][3])
for n,v in by.items():
    v2=sorted(v)
    print(f"{n[:40]:40s} n={len(v):4d} min={v2[0]/1e3:6.2f} p10={v2[len(v2)//10]/1e3:6.2f} p50={v2[len(v2)//2]/1e3:6.2f} p90={v2[len(v2)*9//10]/1e3:6.2f} max={v2[-1]/1e3:6.2f} gap_before_p50={sorted(gaps[n])[len(gaps[n])//2]/1e3 if gaps[n] else 0:6.2f}")
# per-layer pattern for pv: print durations of consecutive pv launches in one token
pv=[d for (n,d,s,e) in tail if n.startswith('mc_attn_pv')]
print('pv seq', [round(x/1e3,1) for x in pv[-64:]])
sc=[d for (n,d,s,e) in tail if n.startswith('mc_attn_scores')]
print('scores seq', [round(x/1e3,1) for x in sc[-64:]])
PY
