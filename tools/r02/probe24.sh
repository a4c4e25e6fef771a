#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
OUT=gpurun_out/r02p24
mkdir -p $OUT
for rep in 1 2; do
  EXPO_HIP_LIB=$R/tools/r02/libs/fused_scalar.so timeout 200 python tools/bench_extra.py > $OUT/head_$rep.json 2>/dev/null
  EXPO_HIP_LIB=$R/tools/r02/libs/disp_lb4.so timeout 200 python tools/bench_extra.py > $OUT/lb4_$rep.json 2>/dev/null
  timeout 200 python tools/bench_extra.py > $OUT/fixup134_$rep.json 2>/dev/null
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r02p24/*_?.json')):
  try:
    d = json.loads(open(f).read().strip().splitlines()[-1])['kernels']
    print(f.split('/')[-1], {k: round(v['ms'] * 1e3, 1) for k, v in d.items() if 'dispatch' in k})
  except Exception as e:
    print(f, 'ERR', e)
PY
