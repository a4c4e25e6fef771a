#!/bin/bash
# r03p41: masked apply backward / vignet backward, 48-byte groups per thread (EXPO_APPLY_GROUPS_PER_THREAD)
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
for g in 1 2 4 8; do
  echo "== groups per thread $g"
  EXPO_APPLY_GROUPS_PER_THREAD=$g timeout 300 python tools/bench_extra.py 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['kernels']
print(' '.join('%s %.1f' % (k, v['ms']*1e3) for k, v in r.items() if 'apply_bwd' in k or 'vignet_apply_bwd' in k))
"
done
