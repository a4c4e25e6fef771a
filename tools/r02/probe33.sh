#!/bin/bash
# fused inference kernel: 86 VGPRs / occupancy 5 (committed) vs launch bounds for 6 and 8 waves per SIMD
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
OUT=gpurun_out/r02p33
mkdir -p $OUT
for rep in 1 2 3; do
  for s in B C; do
    timeout 100 python bench.py --workload infer --shape $s > $OUT/occ5_${s}_$rep.json 2>/dev/null
    EXPO_HIP_LIB=$R/tools/r02/libs/fused_occ6.so timeout 100 python bench.py --workload infer --shape $s > $OUT/occ6_${s}_$rep.json 2>/dev/null
    EXPO_HIP_LIB=$R/tools/r02/libs/fused_occ8.so timeout 100 python bench.py --workload infer --shape $s > $OUT/occ8_${s}_$rep.json 2>/dev/null
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r02p33/*_?_?.json')):
  try:
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split('/')[-1], round(d.get('ms_per_step') * 1e3, 2))
  except Exception as e:
    print(f, 'ERR', e)
PY
EXPO_HIP_LIB=$R/tools/r02/libs/fused_occ8.so timeout 300 python -m pytest tests -m gpu -x -q -k "fused" 2>&1 | tail -2
