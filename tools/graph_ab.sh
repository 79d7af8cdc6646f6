# Runs ON THE GPU BOX: graph replay vs eager launches of the single-GPU headline, alternating on one box
cd /root/repo
for i in 1 2 3; do for A in "" "--no-graph"; do
  python3 bench.py $A --steps 256 --warmup 32 --no-cpu-baseline --no-roofline --no-other-configs 2>/dev/null | tail -n 1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('graph' if d['config']['hipgraph'] else 'eager', round(d['value'],1))"
done; done
