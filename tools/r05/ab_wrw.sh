#!/bin/bash
# A/B of the in-house weight-gradient kernel inside the training iteration (BASELINE config 3), same box, interleaved.
cd "$(dirname "$0")/../.."
for rep in 1 2; do
  for hip in 0 auto 1; do
    echo "== EXPO_HIP_CONV_WRW=$hip (repeat $rep)"
    EXPO_HIP_CONV_WRW=$hip timeout 200 python bench.py --workload train --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
  if l.startswith('{'):
    d=json.loads(l); print('ms_per_iteration %.3f  images/s %.0f' % (d.get('ms_per_step', 0), d.get('value', 0)), d.get('config', {}).get('launches_per_iteration'))
"
  done
done
