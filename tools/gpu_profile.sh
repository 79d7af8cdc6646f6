#!/bin/bash
# Runs IN THE BUILD CONTAINER: stamps the tree's git head into .build_head (git does not travel to the GPU box), then runs
# tools/profile_round.sh there.   usage: tools/gpu_profile.sh r04 [all|stats|traffic|mfma|configs] [timeout s]
R=${1:-r04}; WHAT=${2:-all}; T=${3:-1100}
cd "$(dirname "$0")/.." || exit 1
echo "$(git rev-parse HEAD)$(git diff --quiet || echo +dirty)" > .build_head
/usr/local/graft/bin/gpurun --timeout "$T" -- "bash tools/profile_round.sh $R $WHAT"
