#!/bin/bash
# curve filters: segment-table forward re-evaluation in the masked apply kernels and the fused-penalty backward
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
OUT=gpurun_out/r02p23
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q -k "dispatch or apply or mask or agent or reduction or evaluate" > $OUT/pytest.txt 2>&1
tail -3 $OUT/pytest.txt
for rep in 1 2; do
  EXPO_HIP_LIB=$R/tools/r02/libs/fused_scalar.so timeout 200 python tools/bench_extra.py > $OUT/head_$rep.json 2>/dev/null
  timeout 200 python tools/bench_extra.py > $OUT/new_$rep.json 2>/dev/null
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r02p23/*_?.json')):
  try:
    d = json.loads(open(f).read().strip().splitlines()[-1])['kernels']
    print(f.split('/')[-1], {k: round(v['ms'] * 1e3, 1) for k, v in d.items()})
  except Exception as e:
    print(f, 'ERR', e)
PY
