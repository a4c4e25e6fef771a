#!/bin/bash
# r03p29: critic step with the three critic passes batched into one (default) vs the interpolated third separate
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
timeout 900 python -m pytest tests/test_hip_nets.py tests/test_hip_stats.py -x -q 2>&1 | tail -3
for rep in 1 2 3; do
  for b in 1 0; do
    echo "== batch_critic_passes=$b rep $rep"
    EXPO_BATCH_CRITIC_PASSES=$b timeout 300 python bench.py --workload train --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
  done
done
