#!/bin/bash
# r03p14: register-budget variants. (a) backward kernels with a 5 / 6 waves-per-SIMD budget (Color 125 -> 96 VGPRs + 24 B
# scratch), per-kernel times in the chain; (b) the fused inference kernel at 6 / 8 waves; (c) new gpu tests.
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r03p14
rm -rf $OUT; mkdir -p $OUT
cd $R
Q="--no-cpu-baseline --cold-shape none"
pk() { python -c "
import json,sys
d=json.load(open('$1'))
pk=d['per_kernel']
print('$2: %.4f ms/step' % d['ms_per_step'], ' '.join('%s=%.1f' % (k[4:] if k.startswith('bwd') else 'f'+k[4:], v['ms']*1e3) for k,v in pk.items() if k.startswith('bwd')), '| fwd', ' '.join('%.1f' % (v['ms']*1e3) for k,v in pk.items() if k.startswith('fwd')))
"; }
for rep in 1 2; do
for v in base bwd5 bwd6; do
  lib=$R/tools/r03/libs/$v.so; [ $v = base ] && lib=$R/exposure_amd/libexposure_hip.so
  EXPO_HIP_LIB=$lib python bench.py $Q > $OUT/chain_${v}_$rep.json 2>/dev/null
  pk $OUT/chain_${v}_$rep.json "$v r$rep"
done
done
for g in 1 3 4; do
  EXPO_COLOR_GROUPS_PER_THREAD=$g EXPO_HIP_LIB=$R/tools/r03/libs/bwd5.so python bench.py $Q > $OUT/chain_bwd5_cg$g.json 2>/dev/null
  pk $OUT/chain_bwd5_cg$g.json "bwd5 color groups $g"
done
for v in base fused6 fused8; do
  lib=$R/tools/r03/libs/$v.so; [ $v = base ] && lib=$R/exposure_amd/libexposure_hip.so
  for s in B C; do
    EXPO_HIP_LIB=$lib python bench.py --workload infer --shape $s --steps 50 --warmup 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v infer $s: %.2f us' % (d['ms_per_step']*1e3))"
  done
done
timeout 900 python -m pytest tests/test_hip_vignet.py tests/test_nn_ops.py tests/test_hip_filters.py tests/test_hip_agent.py -x -q -m gpu 2>&1 | tail -4
