#!/bin/bash
# fused inference kernel: scalar fp32 bodies (HEAD) vs pixel-pair bodies (v_pk_*_f32)
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
OUT=gpurun_out/r02p18
mkdir -p $OUT
timeout 300 python -m pytest tests -m gpu -x -q -k "fused or infer or retouch or evaluate or golden" > $OUT/pytest.txt 2>&1
tail -3 $OUT/pytest.txt
for rep in 1 2 3; do
  for s in B C; do
    EXPO_HIP_LIB=$R/tools/r02/libs/fused_scalar.so timeout 100 python bench.py --workload infer --shape $s > $OUT/scalar_${s}_$rep.json 2>/dev/null
    timeout 100 python bench.py --workload infer --shape $s > $OUT/pair_${s}_$rep.json 2>/dev/null
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r02p18/*_?_?.json')):
  try:
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split('/')[-1], d.get('ms_per_step'), d.get('value'))
  except Exception as e:
    print(f, 'ERR', e)
PY
