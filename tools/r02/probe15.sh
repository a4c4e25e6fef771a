#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r02p15
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_hip_agent.py tests/test_hip_reduction.py tests/test_hip_filters.py -x -q 2>&1 | tail -3
for rep in 1 2; do timeout 100 python tools/bench_extra.py > $OUT/extra_$rep.json 2>/dev/null; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02p15/extra_*.json')):
    d=json.load(open(f)); print(f, ' '.join('%s=%.1f' % (k.replace('dispatch','dsp').replace('penalty','pen').replace('apply','ap'), v['ms']*1e3) for k,v in d['kernels'].items()))
PY
