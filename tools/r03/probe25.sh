#!/bin/bash
# r03p25: expo_chain_fused_bwd -- parity tests, first timing, packed-fp16 (default) vs generic curve backward
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
mkdir -p gpurun_out/p25
timeout 900 python -m pytest tests/test_hip_fused_bwd.py -x -q 2>&1 | tail -15
for rep in 1 2; do
  echo "== default"
  timeout 300 python bench.py --workload chain_fused --steps 20 --warmup 5 2>&1 | tail -1 | tee gpurun_out/p25/fused_default_$rep.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print(d['ms_per_step'], c['fused_fwd_ms'], c['fused_bwd_ms'], c['per_step_chain_ms'], c['dparams_max_diff_vs_per_step_rel_to_scale'])"
  echo "== generic curve backward (no packed fp16)"
  EXPO_HIP_LIB=$R/tools/r03/libs/fbwd_nof16x.so timeout 300 python bench.py --workload chain_fused --steps 20 --warmup 5 2>&1 | tail -1 | tee gpurun_out/p25/fused_nof16x_$rep.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print(d['ms_per_step'], c['fused_fwd_ms'], c['fused_bwd_ms'], c['per_step_chain_ms'], c['dparams_max_diff_vs_per_step_rel_to_scale'])"
done
for g in 2 4 16; do
  echo "== groups per thread $g"
  EXPO_FUSED_BWD_GROUPS_PER_THREAD=$g timeout 300 python bench.py --workload chain_fused --steps 20 --warmup 5 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print(d['ms_per_step'], c['fused_fwd_ms'], c['fused_bwd_ms'])"
done
echo "== fp32 storage, shape B"
timeout 300 python bench.py --workload chain_fused --dtype f32 --shape B --steps 20 --warmup 5 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print(d['ms_per_step'], c['fused_fwd_ms'], c['fused_bwd_ms'], c['per_step_chain_ms'])"
timeout 600 python -m pytest tests/test_hip_evaluate.py -x -q 2>&1 | tail -3
