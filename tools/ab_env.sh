#!/bin/bash
# Runs ON THE GPU BOX: alternating bench.py runs of two environments on one box (perf deltas only count same-box, interleaved).
# usage: tools/ab_env.sh <rounds> "<env A>" "<env B>" [bench args...]   -> gpurun_out/ab_env.jsonl (one line per run)
R=$1; A=$2; B=$3; shift 3
mkdir -p gpurun_out
for i in $(seq 1 "$R"); do
  for V in "$A" "$B"; do
    line=$(env $V python3 bench.py --no-cpu-baseline --no-other-configs "$@" 2>/dev/null | tail -n 1)
    echo "{\"env\": \"$V\", \"run\": $i, \"bench\": $line}" >> gpurun_out/ab_env.jsonl
    python3 - "$V" "$line" <<'PY'
import json, sys
d = json.loads(sys.argv[2])
r = d.get("roofline", {})
print(f"{sys.argv[1]:40s} {d['value']:8.1f} tok/s  w13 {r.get('avg_launch_us', 0):.2f} us")
PY
  done
done
