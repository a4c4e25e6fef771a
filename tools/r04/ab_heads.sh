#!/bin/bash
# r04p18: the eight heads' FC layers over packed parameters (one GEMM + one batched GEMM) vs one addmm/lrelu/addmm per head.
OUT=${1:-gpurun_out/r04p18}; mkdir -p $OUT
for i in 1 2 3; do
  for v in 0 1; do
    EXPO_PACKED_HEADS=$v python bench.py --workload train --steps 20 --warmup 3 > $OUT/train_heads${v}_$i.json 2>$OUT/train_heads${v}_$i.err
    python -c "import json; d=json.load(open('$OUT/train_heads${v}_$i.json')); print('EXPO_PACKED_HEADS=$v run $i: %.3f ms' % d['ms_per_step'])"
  done
done
