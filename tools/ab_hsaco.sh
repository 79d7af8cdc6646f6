#!/bin/bash
# Runs ON THE GPU BOX: alternating bench.py runs of several builds of the code object (tools/variants/<name>.hsaco, loaded
# through MC_HSACO) on one box -- perf deltas only count same-box, interleaved.
# usage: tools/ab_hsaco.sh <rounds> "<bench args>" name1 name2 ...   -> gpurun_out/ab_hsaco.jsonl (one line per run)
R=$1; ARGS=$2; shift 2
mkdir -p gpurun_out
for i in $(seq 1 "$R"); do
  for V in "$@"; do
    line=$(MC_HSACO=$PWD/tools/variants/$V.hsaco python3 bench.py --no-cpu-baseline --no-other-configs $ARGS 2>/dev/null | tail -n 1)
    echo "{\"variant\": \"$V\", \"run\": $i, \"bench\": $line}" >> gpurun_out/ab_hsaco.jsonl
    python3 - "$V" "$line" <<'PY'
import json, sys
d = json.loads(sys.argv[2])
r = d.get("roofline", {})
print(f"{sys.argv[1]:24s} {d['value']:8.1f} tok/s  w13 {r.get('avg_launch_us', 0):.2f} us", flush=True)
PY
  done
done
