#!/bin/bash
# groups per thread for the dispatch backward and the masked apply backward (environment knobs, same library)
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
OUT=gpurun_out/r02p43
mkdir -p $OUT
for rep in 1 2; do
  timeout 100 python tools/bench_extra.py > $OUT/base_$rep.json 2>/dev/null
  for g in 1 2; do
    EXPO_DISPATCH_GROUPS_PER_THREAD=$g EXPO_APPLY_GROUPS_PER_THREAD=$g timeout 100 python tools/bench_extra.py > $OUT/g${g}_$rep.json 2>/dev/null
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r02p43/*_?.json')):
  d = json.loads(open(f).read().strip().splitlines()[-1])['kernels']
  print(f.split('/')[-1], {k: round(v['ms'] * 1e3, 1) for k, v in d.items() if 'bwd' in k})
PY
