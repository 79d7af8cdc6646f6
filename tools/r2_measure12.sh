#!/bin/bash
cd /root/repo
O=gpurun_out; L=$O/ab12.log; : > $L
run() { hs=$1; lw=$2; geom=$3; MC_LIN_WAVES=$lw DBGS=0 timeout -k 10 120 python3 tools/gemv_ab.py $hs $geom >> $L 2>> $O/ab12.err || echo "{\"hsaco\": \"$hs\", \"failed\": $?}" >> $L; }
run metalchat_amd/lib/metalchat.hsaco 8 512x1
run tools/variants/xfirst.hsaco 8 512x1
cat $L
MC_HSACO=tools/variants/xfirst_tl.hsaco GEOMS=512x1 CHAIN=4 timeout -k 10 120 python3 tools/lin_timeline.py > $O/tl12.log 2>> $O/tl12.err
python3 - <<'PY'
import json
for l in open('/root/repo/gpurun_out/tl12.log'):
    d=json.loads(l); print(d['which'], 'start', d['start'], 'staged', d['staged'], 'tile0', d['tile_end'][0], 'end', d['end'])
PY
MC_HSACO=tools/variants/xfirst.hsaco timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-other-configs --steps 128 --warmup 16 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench xfirst', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], {k:round(v['avg_launch_us'],2) for k,v in d['roofline']['other_gemvs'].items()})"
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/r02_gputests3.log 2>&1; echo tests rc=$?; tail -3 $O/r02_gputests3.log
