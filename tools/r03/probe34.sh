#!/bin/bash
# r03p34: one-pass backward, per-step sums of the element-wise filters in registers (variant) vs in LDS columns (current)
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
for rep in 1 2 3; do
  for lib in "" tools/r03/libs/fbwd_totreg.so; do
    echo "== lib [$lib] rep $rep"
    EXPO_HIP_LIB=${lib:+$R/$lib} timeout 300 python bench.py --workload chain_fused --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print(d['ms_per_step'], c['fused_fwd_ms'], c['fused_bwd_ms'])"
  done
done
EXPO_HIP_LIB=$R/tools/r03/libs/fbwd_totreg.so timeout 600 python -m pytest tests/test_hip_fused_bwd.py -x -q 2>&1 | tail -2
for lib in "" tools/r03/libs/fbwd_totreg.so; do
  EXPO_HIP_LIB=${lib:+$R/$lib} timeout 300 python bench.py --workload chain_fused --dtype f32 --shape B --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print('f32 B', d['ms_per_step'], c['fused_fwd_ms'], c['fused_bwd_ms'])"
done
