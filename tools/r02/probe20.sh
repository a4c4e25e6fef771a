#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
OUT=gpurun_out/r02p20
mkdir -p $OUT
timeout 300 python tools/r02/probe20.py 64,512,512 > $OUT/split_C.txt 2>&1
tail -12 $OUT/split_C.txt
timeout 200 python tools/r02/probe20.py 16,512,512 > $OUT/split_B.txt 2>&1
tail -9 $OUT/split_B.txt
# fused kernel: scalar / pair+pin
for rep in 1 2; do
  for s in B C; do
    EXPO_HIP_LIB=$R/tools/r02/libs/fused_scalar.so timeout 100 python bench.py --workload infer --shape $s > $OUT/scalar_${s}_$rep.json 2>/dev/null
    timeout 100 python bench.py --workload infer --shape $s > $OUT/pairpin_${s}_$rep.json 2>/dev/null
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r02p20/*_?_?.json')):
  try:
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split('/')[-1], d.get('ms_per_step'), d.get('value'))
  except Exception as e:
    print(f, 'ERR', e)
PY
