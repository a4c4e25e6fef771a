#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r02p12
mkdir -p $OUT
cd $R
B="--no-cpu-baseline --cold-shape none"
for rep in 1 2; do
  timeout 100 python bench.py $B > $OUT/base_C_$rep.json 2>/dev/null
  EXPO_HIP_LIB=$R/tools/r02/libs/stag4.so timeout 100 python bench.py $B > $OUT/stag4_C_$rep.json 2>/dev/null
  EXPO_HIP_LIB=$R/tools/r02/libs/stag16.so timeout 100 python bench.py $B > $OUT/stag16_C_$rep.json 2>/dev/null
done
timeout 100 python bench.py $B --shape A > $OUT/base_A.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02p12/*.json')):
    d=json.load(open(f)); pk=d['per_kernel']
    print('%-16s %.4f ms  %6.0f Mpx/s  fwd_avg %.2f bwd_avg %.2f' % (f.split('/')[-1][:-5], d['ms_per_step'], d['value'], sum(v['ms'] for k,v in pk.items() if k[0]=='f')/8*1e3, sum(v['ms'] for k,v in pk.items() if k[0]=='b')/8*1e3))
PY
