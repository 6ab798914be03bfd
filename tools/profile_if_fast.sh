#!/usr/bin/env bash
# The boxes of the pool differ by +-4 % in sustained clock.  Measure this one first (short headline run); collect the round's
# profile set only on a box at or above the given images/s, so that before / after tables across rounds are comparable.
# usage: tools/profile_if_fast.sh <min images/s> <round script>
set -u
MIN=${1:-22.3}
SCRIPT=${2:-tools/profile_round4.sh}
mkdir -p gpurun_out
V=$(python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['value'])")
echo "box speed: $V images/s (threshold $MIN)" | tee gpurun_out/box_speed.txt
if python -c "import sys; sys.exit(0 if float('$V') >= float('$MIN') else 1)"; then
  bash $SCRIPT > gpurun_out/profile_round.log 2>&1
  echo "profile set collected" | tee -a gpurun_out/box_speed.txt
else
  echo "slow box: nothing collected" | tee -a gpurun_out/box_speed.txt
fi
