#!/bin/bash
# same-box A/B of an environment switch on the optimiser step: tools/r4_ab_env.sh VAR "C2 C3 C5" (each value twice, interleaved)
V=$1; SH=${2:-"C2 C3 C5"}
for rep in 1 2; do for val in 1 0; do for c in $SH; do
  env $V=$val python tools/r4_step_time.py $c 2>/dev/null | tail -1
done; done; done
