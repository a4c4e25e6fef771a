#!/bin/bash
# r03p30: dispatch backward, its two launches on two streams (default for >= 40 MiB tensors) vs back to back
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
for rep in 1 2 3; do
  for st in 2 1; do
    echo "== EXPO_DISPATCH_STREAMS=$st rep $rep"
    EXPO_DISPATCH_STREAMS=$st timeout 300 python tools/bench_extra.py 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['kernels']
for k in ('dispatch_fwd+penalty','dispatch_bwd+penalty'):
  print(k, r[k])
"
  done
done
