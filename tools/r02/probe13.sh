#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r02p13
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_hip_agent.py tests/test_hip_evaluate.py -x -q 2>&1 | tail -3
for rep in 1 2 3; do timeout 100 python bench.py --workload infer --shape B > $OUT/infer_B_$rep.json 2>/dev/null; done
timeout 100 python bench.py --workload infer --shape C > $OUT/infer_C.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02p13/*.json')):
    d=json.load(open(f)); print(f, d['ms_per_step'], d['config']['max_abs_diff_fused_vs_per_step'])
PY
