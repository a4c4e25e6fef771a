#!/bin/bash
# r03p16: fused inference kernel, two steps per loop trip with ping-pong pixel arrays (no loop-carried copies):
# old (86 VGPRs, 5 waves) vs new (111 VGPRs, 4 waves) vs new with a 5-wave register budget; + dispatch ids sorted vs cycling
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r03p16
rm -rf $OUT; mkdir -p $OUT
cd $R
for rep in 1 2; do
for v in fused_old base fusedpp5; do
  lib=$R/tools/r03/libs/$v.so; [ $v = base ] && lib=$R/exposure_amd/libexposure_hip.so
  for s in B C; do
    EXPO_HIP_LIB=$lib python bench.py --workload infer --shape $s --steps 50 --warmup 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v infer $s rep$rep: %.2f us  (diff vs per-step %.2e)' % (d['ms_per_step']*1e3, d['config']['max_abs_diff_fused_vs_per_step']))"
  done
done
done
for ids in cycle sorted; do
  python tools/bench_extra.py --ids $ids | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$ids', {k: round(v['ms']*1e3,1) for k,v in d['kernels'].items() if k.startswith('dispatch')})"
done
timeout 900 python -m pytest tests/test_hip_agent.py -x -q -m gpu -k "fused" 2>&1 | tail -3
