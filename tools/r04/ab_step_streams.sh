#!/bin/bash
# r04p16: the training steps' independent branches on a second stream vs one stream; which step kind gains or loses.
OUT=${1:-gpurun_out/r04p16}; mkdir -p $OUT
run() { tag=$1; shift; g=on; case $tag in *eager*) g=off;; esac
  env "$@" python bench.py --workload train --steps 20 --warmup 3 --graph $g > $OUT/train_$tag.json 2>$OUT/train_$tag.err
  python -c "import json; d=json.load(open('$OUT/train_$tag.json')); print('$tag: %.3f ms' % d['ms_per_step'])"; }
for i in 1 2; do
  run one_$i EXPO_STEP_STREAMS=1
  run critic_only_$i EXPO_STEP_STREAMS_WHERE=c
  run generator_only_$i EXPO_STEP_STREAMS_WHERE=g
  run both_$i EXPO_STEP_STREAMS_WHERE=gc
done
run one_eager EXPO_STEP_STREAMS=1
run both_eager EXPO_STEP_STREAMS_WHERE=gc
