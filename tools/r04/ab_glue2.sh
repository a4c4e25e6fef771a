#!/bin/bash
# r04p21: convnet input assembly (expo_planes_concat) and the G step's loss glue (expo_generator_losses) on / off.
OUT=${1:-gpurun_out/r04p21}; mkdir -p $OUT
run() { tag=$1; shift; env "$@" python bench.py --workload train --steps 20 --warmup 3 > $OUT/train_$tag.json 2>$OUT/train_$tag.err
  python -c "import json; d=json.load(open('$OUT/train_$tag.json')); print('$tag: %.3f ms' % d['ms_per_step'])"; }
for i in 1 2 3; do
  run both_off_$i EXPO_PLANES_CONCAT=0 EXPO_FUSED_G_LOSSES=0
  run concat_only_$i EXPO_FUSED_G_LOSSES=0
  run gloss_only_$i EXPO_PLANES_CONCAT=0
  run both_on_$i EXPO_X=0
done
