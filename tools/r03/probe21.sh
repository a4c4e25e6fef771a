#!/bin/bash
# r03p21: Color / Tone backward groups per thread with the current kernels (env only)
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
Q="--no-cpu-baseline --cold-shape none"
for rep in 1 2; do
for g in 1 2 3 4; do
  EXPO_COLOR_GROUPS_PER_THREAD=$g EXPO_TONE_GROUPS_PER_THREAD=$g python bench.py $Q > /tmp/b.json 2>/dev/null
  python -c "
import json; d=json.load(open('/tmp/b.json')); pk=d['per_kernel']; r=d['roofline']
print('groups $g rep $rep: step %.4f | bwd C %.1f T %.1f (raw pairs; overhead %.2f us)' % (d['ms_per_step'], pk['bwd_C']['ms']*1e3, pk['bwd_T']['ms']*1e3, r['event_pair_overhead_ms']*1e3))"
done
done
