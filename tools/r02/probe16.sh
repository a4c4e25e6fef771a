#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r02p16
mkdir -p $OUT
cd $R
export EXPO_HIP_LIB=$R/tools/r02/libs/single.so
for rep in 1 2; do
  EXPO_DISPATCH_SINGLE=0 timeout 100 python tools/bench_extra.py > $OUT/two_$rep.json 2>/dev/null
  EXPO_DISPATCH_SINGLE=1 timeout 100 python tools/bench_extra.py > $OUT/single_$rep.json 2>/dev/null
done
EXPO_DISPATCH_SINGLE=1 timeout 300 python -m pytest tests/test_hip_agent.py tests/test_hip_reduction.py -x -q 2>&1 | tail -2
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02p16/*.json')):
    d=json.load(open(f)); print(f, ' '.join('%s=%.1f' % (k.replace('dispatch','dsp').replace('penalty','pen'), v['ms']*1e3) for k,v in d['kernels'].items() if 'dispatch' in k))
PY
