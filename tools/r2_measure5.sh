#!/bin/bash
cd /root/repo
O=gpurun_out; : > $O/tl5.log
for v in tl2 tl2_stream tl2_noload; do
  echo "== $v" >> $O/tl5.log
  MC_HSACO=tools/variants/$v.hsaco GEOMS=512x1 timeout -k 10 120 python3 tools/lin_timeline.py >> $O/tl5.log 2>> $O/tl5.err
done
python3 - <<'PY'
import json
for l in open('/root/repo/gpurun_out/tl5.log'):
    if l.startswith('=='): print(l.strip()); continue
    d=json.loads(l); print(d['which'], 'clock', d['clock_mhz'], 'end', d['end'], 'staged', d['staged'])
PY
