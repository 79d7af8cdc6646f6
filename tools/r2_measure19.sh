#!/bin/bash
cd /root/repo
O=gpurun_out
j() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['value'],1), 'tok/s', round(d['ms_per_step'],4), 'ms', d['config'].get('parallelism'))"; }
timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-other-configs --no-roofline --steps 128 --warmup 16 2>>$O/m19.err | j n1
timeout -k 10 300 python3 bench.py --gpus 2 --share-device --no-cpu-baseline --no-other-configs --no-roofline --steps 128 --warmup 16 2>>$O/m19.err | j n2share
timeout -k 10 300 python3 bench.py --gpus 4 --share-device --no-cpu-baseline --no-other-configs --no-roofline --steps 128 --warmup 16 2>>$O/m19.err | j n4share
timeout -k 10 300 python3 bench.py --model llama3.2-1b --wbits 16 --no-cpu-baseline --no-other-configs --steps 256 --warmup 32 2>>$O/m19.err > $O/l32_1b.json; python3 - <<'PY'
import json
d=json.load(open('/root/repo/gpurun_out/l32_1b.json'))
print('llama3.2-1b bf16', round(d['value'],1), 'tok/s', d['ms_per_step'], 'ms/token; per layer us ~', round((d['ms_per_step']*1e3-60)/16,1), json.dumps(d.get('roofline',{}).get('other_gemvs',{}))[:400], d.get('roofline',{}).get('avg_launch_us'))
PY
