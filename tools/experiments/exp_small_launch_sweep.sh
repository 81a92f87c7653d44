#!/bin/bash
# Work-unit size (PT355_EXP_UNIT = log2 rays) x dry-region peek (PT355_EXP_PEEK) over launch sets of 1..32 frames: the data behind
# trace_launch_shape() in pt355.hip.  GPU box; one process per setting.
LIB=${1:-$PWD/pathtracer_amd/libpt355.so}
export PT355_LIB=$LIB
for res in "1280 720" "1920 1080"; do
for peek in 0 1; do for unit in 6 7 8; do
  export PT355_EXP_PEEK=$peek PT355_EXP_UNIT=$unit
  timeout -k 10 100 python -u tools/experiments/exp_small_launch.py $res || exit 1
done; done; done
