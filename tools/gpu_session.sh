#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): a list of steps, each under its own timeout; a step that is killed (rc >= 124) ends the
# session (no further GPU step behind a hung one), a step that merely fails (tests: rc 1) does not.
# usage: tools/gpu_session.sh "<timeout s>|<log name>|<command>" ...
mkdir -p gpurun_out
for step in "$@"; do
  IFS='|' read -r T LOG CMD <<< "$step"
  echo "=== [$LOG] $CMD" | tee -a gpurun_out/session.log
  timeout -k 10 "$T" bash -c "$CMD" > "gpurun_out/$LOG" 2>&1
  rc=$?
  echo "=== [$LOG] rc=$rc" | tee -a gpurun_out/session.log
  tail -n 6 "gpurun_out/$LOG"
  if [ $rc -ge 124 ]; then echo "step killed: stopping the session"; exit $rc; fi
done
exit 0
