#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r02p11
mkdir -p $OUT
cd $R
B="--no-cpu-baseline --cold-shape none --shape A"
for rep in 1 2 3; do
  (cd tools/r02/old && timeout 100 python bench.py --no-cpu-baseline --shape A > $OUT/old_A_$rep.json 2>/dev/null)
  timeout 100 python bench.py $B > $OUT/new_A_$rep.json 2>/dev/null
  EXPO_CHAIN_SNAKE=0 timeout 100 python bench.py $B > $OUT/nosnake_A_$rep.json 2>/dev/null
  EXPO_HIP_LIB=$R/tools/r02/libs/earlyp.so timeout 100 python bench.py $B > $OUT/earlyp_A_$rep.json 2>/dev/null
  timeout 100 python bench.py $B --graph off > $OUT/eager_A_$rep.json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02p11/*.json')):
    d=json.load(open(f)); pk=d['per_kernel']
    print('%-16s %.4f ms  %6.0f Mpx/s  fwd_avg %.2f bwd_avg %.2f' % (f.split('/')[-1][:-5], d['ms_per_step'], d['value'], sum(v['ms'] for k,v in pk.items() if k[0]=='f')/8*1e3, sum(v['ms'] for k,v in pk.items() if k[0]=='b')/8*1e3))
PY
