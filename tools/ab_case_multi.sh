#!/bin/bash
# Runs ON THE GPU BOX: alternating tools/configs_run.py runs of ONE case under several environments on one box.
# usage: tools/ab_case_multi.sh <rounds> <case substring> "<env 1>" "<env 2>" ...   -> stdout (and gpurun_out/ab_case.jsonl)
R=$1; CASE=$2; shift 2
mkdir -p gpurun_out
for i in $(seq 1 "$R"); do
  for V in "$@"; do
    line=$(env CASE="$CASE" $V python3 tools/configs_run.py 2>/dev/null | tail -n 1)
    echo "{\"env\": \"$V\", \"run\": $i, \"result\": $line}" >> gpurun_out/ab_case.jsonl
    echo "$V  $line"
  done
done
