#!/bin/bash
# r03p42: mask kernels with the incremental row / column walk (PixelWalk) vs a row / column computation per pixel
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
timeout 900 python -m pytest tests/test_hip_filters.py tests/test_hip_vignet.py tests/test_hip_agent.py -x -q -k "mask or vignet or apply" 2>&1 | tail -3
for rep in 1 2; do
  for lib in "" tools/r03/libs/walk_old.so; do
    echo "== lib [$lib] rep $rep"
    EXPO_HIP_LIB=${lib:+$R/$lib} timeout 300 python tools/bench_extra.py 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['kernels']
print(' '.join('%s %.1f' % (k, v['ms']*1e3) for k, v in r.items() if 'apply' in k or 'vignet' in k))
"
  done
done
