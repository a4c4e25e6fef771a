#!/bin/bash
# r04p17: one-launch Adam (expo_adam_step) vs torch's fused multi-tensor Adam in the training iteration.
OUT=${1:-gpurun_out/r04p17}; mkdir -p $OUT
for i in 1 2 3; do
  for v in 0 1; do
    EXPO_HIP_ADAM=$v python bench.py --workload train --steps 20 --warmup 3 > $OUT/train_adam${v}_$i.json 2>$OUT/train_adam${v}_$i.err
    python -c "import json; d=json.load(open('$OUT/train_adam${v}_$i.json')); print('EXPO_HIP_ADAM=$v run $i: %.3f ms' % d['ms_per_step'])"
  done
done
