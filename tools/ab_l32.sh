#!/bin/bash
# Runs ON THE GPU BOX: Llama-3.2-1B (the reference's default model, bf16 weights) under several environments, alternating on one box.
# usage: tools/ab_l32.sh <rounds> "<env 1>" "<env 2>" ...
R=$1; shift
for i in $(seq 1 "$R"); do
  for V in "$@"; do
    line=$(env $V python3 bench.py --model llama3.2-1b --wbits 16 --steps 256 --warmup 32 --no-cpu-baseline --no-other-configs --no-roofline 2>/dev/null | tail -n 1)
    python3 - "$V" "$line" <<'PY'
import json, sys
d = json.loads(sys.argv[2])
print(f"{sys.argv[1]:24s} llama3.2-1b bf16 {d['value']:8.1f} tokens/s", flush=True)
PY
  done
done
