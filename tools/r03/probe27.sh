#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
mkdir -p gpurun_out/p27
timeout 900 python -m pytest tests/test_hip_fused_bwd.py -x -q 2>&1 | tail -8 | cut -c1-250
for rep in 1 2; do
timeout 300 python bench.py --workload chain_fused --steps 20 --warmup 5 2>&1 | tail -1 | tee gpurun_out/p27/fused_$rep.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print(d['ms_per_step'], d['value'], c['fused_fwd_ms'], c['fused_bwd_ms'], c['per_step_chain_ms'], c['dparams_max_diff_vs_per_step_rel_to_scale'])"
done
timeout 300 python bench.py --workload chain_fused --shape B --steps 20 --warmup 5 2>&1 | tail -1 | tee gpurun_out/p27/fused_B.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print(d['ms_per_step'], d['value'], c['fused_fwd_ms'], c['fused_bwd_ms'], c['per_step_chain_ms'])"
timeout 300 python bench.py --workload chain_fused --shape A --steps 50 --warmup 5 2>&1 | tail -1 | tee gpurun_out/p27/fused_A.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print(d['ms_per_step'], d['value'], c['fused_fwd_ms'], c['fused_bwd_ms'], c['per_step_chain_ms'])"
