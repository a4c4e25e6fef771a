#!/bin/bash
# A/B of the in-house convolution kernels inside the training iteration (BASELINE config 3), same box, interleaved.
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
for rep in 1 2; do
  for hip in 0 1; do
    echo "== EXPO_HIP_CONV=$hip (repeat $rep)"
    EXPO_HIP_CONV=$hip EXPO_HIP_CONV_BWD=$hip timeout 300 python bench.py --workload train --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
  if l.startswith('{'):
    d=json.loads(l); print('ms_per_iteration %.3f  images/s %.0f' % (d.get('ms_per_step', d.get('value')), d.get('value', 0)) , {k: d[k] for k in ('metric','unit') if k in d})
"
  done
done
