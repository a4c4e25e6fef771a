#!/bin/bash
# r04: is the HBM-cold "outlier" (fwd E 141 us vs 122) a property of the FILTER or of the POSITION in the sequence?
R=${GRAFT_REPO_ROOT:-$PWD}
Q="--no-cpu-baseline --cold-shape none --no-legs --steps 4 --warmup 2 --kernel-reps 12"
for order in 0,1,2,3,4,5,6,7 1,0,2,3,4,5,6,7 2,1,0,3,4,5,6,7 6,1,2,3,4,5,0,7 0,1,2,3,4,7,6,5; do
  python $R/bench.py --shape 256,512,512 $Q --order $order 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); pk=d['per_kernel']
print('order $order: ' + ' '.join('%s %.1f' % (k, v['ms']*1e3) for k, v in pk.items()))"
done
